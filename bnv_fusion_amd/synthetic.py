"""Synthetic benchmark frames (SURVEY.md section 8d / BASELINE.md section 3).

Host-side data generation only (numpy, float64 like the reference's dataset code): depth image ->
camera points + Sobel normals -> world frame -> ``input_pts [1, N, 6]``.  The arithmetic follows
FusionInferenceAbstractDataset.__getitem__ (fusion_inference_dataset.py:40-90) with the kornia
0.6.2 normals it calls restated as in geometry.py:515-527.
"""
import math

import numpy as np

INTRINSICS = np.array([[525.0, 0.0, 319.5], [0.0, 525.0, 239.5], [0.0, 0.0, 1.0]])
# volume dimensions giving exactly 128^3 (v=0.02) / 256^3 (v=0.01) / 512^3 (v=0.01) grids
GRID_DIMS = {128: (2.52, 0.02), 256: (2.54, 0.01), 512: (5.10, 0.01), 64: (1.24, 0.02)}


def depth_image(t, H=480, W=640, seed=0):
    """depth(u, v) = 1.5 + 0.2 sin(u/40) cos(v/30) + N(0, 0.002) m, stored as uint16 millimetres."""
    rng = np.random.default_rng(seed + 1000 * t)
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    d = 1.5 + 0.2 * np.sin(u / 40.0) * np.cos(v / 30.0) + rng.normal(0.0, 0.002, size=(H, W))
    return np.round(d * 1000.0).astype(np.uint16).astype(np.float64) / 1000.0


def pose(t):
    """T_wc(t) = translate(0, 0, -1.5) . R_y(0.5 deg * t)."""
    a = math.radians(0.5 * t)
    T = np.eye(4)
    T[:3, :3] = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    T[:3, 3] = [0.0, 0.0, -1.5]
    return T


def depth_to_input_pts(depth, intr, T_wc, max_depth=10.0):
    depth = np.asarray(depth, dtype=np.float64)
    H, W = depth.shape
    fx, fy, cx, cy = intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2]
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    xyz = np.stack([(u - cx) / fx * depth, (v - cy) / fy * depth, depth], axis=0)
    p = np.pad(xyz, ((0, 0), (1, 1), (1, 1)), mode="edge")
    gx = (p[:, :-2, 2:] + 2 * p[:, 1:-1, 2:] + p[:, 2:, 2:]
          - p[:, :-2, :-2] - 2 * p[:, 1:-1, :-2] - p[:, 2:, :-2]) / 8.0
    gy = (p[:, 2:, :-2] + 2 * p[:, 2:, 1:-1] + p[:, 2:, 2:]
          - p[:, :-2, :-2] - 2 * p[:, :-2, 1:-1] - p[:, :-2, 2:]) / 8.0
    n = np.cross(gx, gy, axis=0)
    n = n / np.maximum(np.linalg.norm(n, axis=0, keepdims=True), 1e-12)
    mask = (depth > 0) & (depth < max_depth)
    R, tr = T_wc[:3, :3], T_wc[:3, 3]
    pts_w = xyz.reshape(3, -1).T @ R.T + tr
    nrm_w = n.reshape(3, -1).T @ R.T
    return np.concatenate([pts_w, nrm_w], axis=-1)[mask.reshape(-1)]


def frame(t, H=480, W=640, seed=0):
    """-> numpy float32 [1, N, 6] (the float64 -> .float() cast of run_e2e.py:249)."""
    intr = INTRINSICS.copy()
    if (H, W) != (480, 640):
        intr[0] *= W / 640.0
        intr[1] *= H / 480.0
    pts = depth_to_input_pts(depth_image(t, H, W, seed), intr, pose(t))
    return pts.astype(np.float32)[None]
