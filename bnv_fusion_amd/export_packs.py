"""Writes what a host that is NOT Python needs to drive the C ABI (examples/capi_host.cpp): the packed weights the
library's kernels read (the same arrays the Python side uploads; layouts: weights.py / DESIGN.md section 2), the
volume's grid description as the bytes of ``bnv_grid_t``, the camera, and a few frames.

    python -m bnv_fusion_amd.export_packs OUT_DIR [--frames 6] [--hw 240 320] [--grid 256] [--checkpoint tcnn]

Files: meta.bin (int32 H, W, n_frames, mlp_mode | float64 max_depth | float64 K[9] | bnv_grid_t | pad to 8 |
float64 T_wc[16] per frame), pointnet_pack.bin, sdfmlp_pack.bin (raw little-endian float32), depth_<k>.u16 (raw
uint16 millimetres, row-major).
"""
import argparse
import os
import struct

import numpy as np


def export(out_dir, model, volume, depths_u16, K, poses, max_depth=3.0, mlp_mode=None):
    """``model``: a LitFusionPointNet of this package (its packs are read back from the device), ``volume``: the
    SparseVolume whose grid the frames are fused into."""
    from . import _lib
    os.makedirs(out_dir, exist_ok=True)
    model.pointnet_pack.detach().cpu().numpy().astype("<f4").tofile(os.path.join(out_dir, "pointnet_pack.bin"))
    model.nerf.sdf_pack.detach().cpu().numpy().astype("<f4").tofile(os.path.join(out_dir, "sdfmlp_pack.bin"))
    grid, _ = model._grid(volume.n_xyz, volume.min_coords, volume.max_coords, volume.voxel_size)
    depths = [np.ascontiguousarray(d, dtype="<u2") for d in depths_u16]
    H, W = depths[0].shape
    if mlp_mode is None:
        mlp_mode = _lib.model_mode(model)
    gb = bytes(_lib.grid_with_mode(grid, mlp_mode))
    head = struct.pack("<4i d 9d", H, W, len(depths), int(mlp_mode), float(max_depth),
                       *np.asarray(K, dtype=np.float64)[:3, :3].reshape(-1))
    body = head + gb
    body += b"\0" * (-len(body) % 8)
    for T in poses:
        body += np.asarray(T, dtype="<f8").reshape(16).tobytes()
    with open(os.path.join(out_dir, "meta.bin"), "wb") as fh:
        fh.write(body)
    for k, d in enumerate(depths):
        d.tofile(os.path.join(out_dir, f"depth_{k}.u16"))
    return out_dir


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("out_dir")
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--hw", type=int, nargs=2, default=[240, 320])
    ap.add_argument("--grid", type=int, default=256, choices=[128, 256, 512])
    ap.add_argument("--checkpoint", default="fp32", choices=["fp32", "tcnn"])
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()
    import bnv_fusion_amd as bnv
    from . import synthetic
    dims, voxel = synthetic.GRID_DIMS[a.grid]
    model = bnv.load_pretrained(device=a.device, voxel_size=voxel, tiny_cuda=a.checkpoint == "tcnn")
    vol = bnv.SparseVolume(8, voxel, np.array([dims] * 3), 8, device=a.device)
    H, W = a.hw
    export(a.out_dir, model, vol, [synthetic.depth_u16(t, H, W) for t in range(a.frames)], synthetic.intrinsics(H, W),
           [synthetic.pose(t) for t in range(a.frames)])
    print(a.out_dir, sorted(os.listdir(a.out_dir)))


if __name__ == "__main__":
    main()
