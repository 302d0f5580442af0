"""Golden vector for the depth front end's POINTS, captured from the reference's own
functions.

Build-container only (needs /root/reference):  python tests/golden/make_golden_frontend.py
FusionInferenceAbstractDataset.__getitem__ (src/datasets/fusion_inference_dataset.py:67-75) builds the world points
with ``geometry.depth2xyz`` and ``geometry.get_homogeneous``; those two functions are called here exactly as that
method calls them (the method itself needs image files, cv2 and kornia).  The reference pins numpy 1.x, where
``float32_array - float64_scalar`` stays float32 (value-based casting); under this container's numpy 2 the same
source would promote to float64, so the intrinsics are handed over as float32, which selects the float32 loops the
reference's environment selects.  The normals (kornia.depth_to_normals) cannot be captured: kornia is absent.

Only DATA is written (tests/golden/frontend_120.npz).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_shims  # noqa: E402


def main():
    ref_shims.install()
    from src.utils import geometry
    from oracle import bnv_oracle as orc
    H, W = 120, 160
    intr = orc.SYNTHETIC_INTRINSICS.copy()
    intr[:2] *= 0.25
    depth = orc.synthetic_depth(5, H, W)
    depth[10:14, 20:40] = 0.0                      # invalid pixels
    depth[50, 60] = 12.0                           # beyond max_depth
    T_wc = orc.synthetic_pose(5)
    max_depth = 10.0
    mask = np.logical_and(depth > 0, depth < max_depth)        # common.py:110-113
    d = depth * mask
    pts_c = geometry.depth2xyz(d, intr.astype(np.float32)).reshape(-1, 3)               # :67
    pts_w = (T_wc @ geometry.get_homogeneous(pts_c).T)[:3, :].T                         # :68
    np.savez_compressed(os.path.join(HERE, "frontend_120.npz"), depth=depth, intr=intr, T_wc=T_wc,
                        max_depth=max_depth, pts_w=pts_w[mask.reshape(-1)], n_valid=int(mask.sum()))
    print("frontend_120:", pts_w.shape, "valid", int(mask.sum()), pts_w.dtype)


if __name__ == "__main__":
    main()
