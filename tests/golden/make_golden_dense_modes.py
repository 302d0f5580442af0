"""Golden vectors from the reference for the two other branches of decode_feature_grid_w_pts
(local_point_fusion.py:265-367): interpolate_decode=False (nearest voxel, one evaluation) and global_coords=True (the
signature default: trilinear features, the MLP on coords / (res - 1), unscaled).

Build-container only (needs /root/reference):  python tests/golden/make_golden_dense_modes.py
Inputs are the points of dense_decode_64.npz (the dense grids come from encode_pointcloud(return_dense=True) on
them); only DATA is written: queries + the reference's (sdf, neighbor_feats) per branch -> dense_modes_64.npz.
Queries include exact integers and .5 ties (torch.round is half-to-even), points off the surface (invalid), points
outside the grid on either side (zero padding) and points on the last voxel.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_shims  # noqa: E402


def main():
    torch.set_num_threads(8)
    z = np.load(os.path.join(HERE, "dense_decode_64.npz"))
    voxel = float(z["voxel_size"])
    model, SV = ref_shims.build_reference_model(voxel, "/tmp/refwork")
    vol = SV(8, voxel, z["dims"], 8, device="cpu")
    from src.utils import voxel_utils as ref_vu
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        p = torch.from_numpy(z["input_pts"])
        fg, mask, uids, _ = model.encode_pointcloud(p.clone(), vol.n_xyz, vol.min_coords, vol.max_coords, voxel,
                                                    return_dense=True)
        u3 = ref_vu.unflatten(uids[mask[0, 0].reshape(-1)[uids] >= 8], vol.n_xyz).float()
        q = u3[torch.randint(len(u3), (900,), generator=g)] + (torch.rand(900, 3, generator=g) - 0.5) * 1.2
        q[:120] = torch.round(q[:120] * 2) / 2                       # integers and .5 ties
        q[120:220] += (torch.rand(100, 3, generator=g) - 0.5) * 6    # off the surface -> invalid
        q[220:260] = torch.rand(40, 3, generator=g) * 70 - 3.5       # some outside the grid (zero padding)
        q[260:270] = torch.tensor([63.0, 63.0, 63.0]) - torch.rand(10, 3, generator=g) * 0.6
        q = q[None]
        out = {}
        sdf, nf = model.decode_feature_grid_w_pts(q, fg, mask, voxel, vol.min_coords, global_coords=True)
        out["sdf_global"], out["feats_global"] = sdf.numpy(), nf.numpy()
        assert model.interpolate_decode
        model.interpolate_decode = False
        sdf, nf = model.decode_feature_grid_w_pts(q, fg, mask, voxel, vol.min_coords, global_coords=False)
        model.interpolate_decode = True
        out["sdf_nearest"], out["feats_nearest"] = sdf.numpy(), nf.numpy()
    for k in ("sdf_global", "sdf_nearest"):
        print(k, out[k].shape, "valid frac", float((out[k] != np.float32(voxel)).mean()),
              "range", float(out[k].min()), float(out[k].max()))
    np.savez_compressed(os.path.join(HERE, "dense_modes_64.npz"), queries=q.numpy(), voxel_size=voxel, dims=z["dims"],
                        **out)


if __name__ == "__main__":
    main()
