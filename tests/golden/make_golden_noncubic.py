"""Golden vectors from the reference itself for a configuration none of the other fixtures has: a NON-CUBIC volume
(dimensions 2.06 x 1.30 x 2.54 m -> n_xyz differs on every axis, so every flatten / unflatten / neighbour / brick index
is exercised with three different strides), voxel 0.02, ``min_pts_in_grid`` 5 instead of the default 8, and a scene
that sticks out of the volume on two axes (the bounds mask cuts on y and x).

Build-container only (needs /root/reference):  python tests/golden/make_golden_noncubic.py
12 frames (320x240) of the bench's synthetic pan: per frame encode_pointcloud(return_dense=False) -> track_n_pts ->
_integrate (run_e2e.py:83-98); at the end SparseVolume.decode_pts of the 3x3x3 lattice of 384 voxels of the last
frame (sparse_volume.py:717-738).  Only DATA is written (tests/golden/noncubic.npz).
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_shims  # noqa: E402

OFF27 = np.array([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)], dtype=np.int64)
H, W = 240, 320
FRAMES = [3 * k for k in range(12)]
DIMS = np.array([2.06, 1.30, 2.54])
VOXEL = 0.02
MIN_PTS = 5


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    torch.set_num_threads(8)
    from bnv_fusion_amd import synthetic
    ref_shims.install()
    cfg = ref_shims.make_cfg(VOXEL)
    cfg["model"]["min_pts_in_grid"] = MIN_PTS
    cwd = os.getcwd()
    os.makedirs("/tmp/refwork", exist_ok=True)
    os.chdir("/tmp/refwork")
    try:
        from src.models.fusion.local_point_fusion import LitFusionPointNet
        from src.models.sparse_volume import SparseVolume as SV
        model = LitFusionPointNet(cfg)
    finally:
        os.chdir(cwd)
    sd = ref_shims.load_checkpoint_state_dict(os.path.join(ref_shims.REFERENCE_ROOT, "pretrained", "pointnet.ckpt"))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    model.eval()
    model.freeze()
    assert model.min_pts_in_grid == MIN_PTS
    vol = SV(8, VOXEL, DIMS, MIN_PTS, device="cpu")
    print("n_xyz", vol.n_xyz.tolist(), "min", vol.min_coords.tolist(), "max", vol.max_coords.tolist(), flush=True)
    assert len(set(vol.n_xyz.tolist())) == 3
    K = synthetic.intrinsics(H, W)
    out = {"voxel_size": VOXEL, "dims": DIMS, "min_pts": MIN_PTS, "frames": np.asarray(FRAMES), "hw": np.asarray([H, W]),
           "n_xyz": np.asarray(vol.n_xyz.tolist()), "feature_stride": 8}
    n_avg, n_out, hashes, pts_sha, inside = [], [], [], [], []
    last = None
    for k, t in enumerate(FRAMES):
        t0 = time.time()
        d16 = synthetic.depth_u16(t, H, W)
        T = synthetic.pose(t)
        pts = synthetic.depth_to_input_pts(d16.astype(np.float64) / 1000.0, K, T, max_depth=3.0).astype(np.float32)[None]
        pts_sha.append(sha(pts))
        p = pts[0, :, :3]
        ok = np.isfinite(p).all(1)
        inside.append(float(((p[ok] > vol.min_coords.numpy()) & (p[ok] < vol.max_coords.numpy())).all(1).mean()))
        with torch.no_grad():
            f, c, ids, g, n = model.encode_pointcloud(torch.from_numpy(pts).clone(), vol.n_xyz, vol.min_coords,
                                                      vol.max_coords, vol.voxel_size, return_dense=False)
            vol.track_n_pts(n)
            model._integrate(vol, g, f, c)
        ids_h, c_h = ids.numpy().astype(np.int64), c.numpy().reshape(-1).astype(np.int64)
        hashes.append(sha(ids_h) + sha(c_h))
        out[f"flat_ids_delta_{k}"] = np.diff(ids_h, prepend=0).astype(np.int32)
        out[f"pcounts_{k}"] = c_h.astype(np.int16)
        if k % 4 == 0 or k == len(FRAMES) - 1:
            out[f"feats8_{k}"] = f.numpy()[::8].copy()
        n_avg.append(float(n))
        n_out.append(len(ids_h))
        last = g
        print(f"frame {t}: {int(ok.sum())} points, {100 * inside[-1]:.0f} % inside, {len(ids_h)} voxels, min count "
              f"{int(c_h.min())}, n_avg {float(n):.3f}, {time.time() - t0:.1f}s", flush=True)
    out["n_avg_pts"] = np.asarray(n_avg, dtype=np.float32)
    out["n_out"] = np.asarray(n_out)
    out["ids_counts_sha256"] = np.asarray(hashes)
    out["input_pts_sha256"] = np.asarray(pts_sha)
    out["inside_fraction"] = np.asarray(inside)
    vol.to_tensor()
    keys = vol.active_coordinates.numpy()
    out["volume_keys"] = keys.astype(np.int16)                       # insertion order
    out["volume_weights"] = vol.weights.numpy().reshape(-1).copy()
    out["volume_feats8"] = vol.features.numpy()[::8].copy()
    w = vol.weights.numpy().reshape(-1)
    print("volume rows", len(keys), "weights in [5, 8):", int(((w >= 5) & (w < 8)).sum()), "weights < 5:",
          int((w < 5).sum()), flush=True)
    g = last.numpy()
    n = len(g)
    starts = [int(s) for s in np.linspace(n * 0.05, n * 0.95 - 64, 6)]
    origins = np.concatenate([g[s: s + 64] for s in starts])
    r = np.arange(0, 1.5, 0.5) - 0.5                                   # sparse_volume.py:717-720
    lat = np.stack(np.meshgrid(r, r, r, indexing="ij"), -1).reshape(27, 3)
    sdf = []
    with torch.no_grad():
        for b0 in range(0, len(origins), 128):
            o = origins[b0: b0 + 128]
            vc = torch.from_numpy((o[:, None, :] + lat[None]).astype(np.float32))[None]
            sdf.append(vol.decode_pts(vc, model.nerf, None, is_coords=True, query_tensor=False)[0, :, :, 0].numpy())
    sdf = np.concatenate(sdf)
    out["decode_origins"] = origins.astype(np.int16)
    out["decode_sdf"] = sdf
    nbr = np.unique((origins[:, None, :] + OFF27[None]).reshape(-1, 3), axis=0)
    nf, nw, _ = vol.query(torch.from_numpy(nbr))
    present = nw.numpy().reshape(-1) > 0
    out["nbr_keys"] = nbr[present].astype(np.int16)
    out["nbr_feats"] = nf.numpy()[present]
    out["nbr_weights"] = nw.numpy().reshape(-1)[present]
    print("decode: live fraction", float((sdf != np.float32(VOXEL)).mean()), flush=True)
    path = os.path.join(HERE, "noncubic.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
