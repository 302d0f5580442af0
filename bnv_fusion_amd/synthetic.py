"""Synthetic benchmark frames (SURVEY.md section 8d / BASELINE.md section 3).

Deviation from the letter of BASELINE.md section 3, in the direction of its stated intent ("frames overlap
as in real sequences", "decode masks are live"): there the depth pattern was a function of the pixel
only, i.e. glued to the rotating camera, so the observed "scene" moved >1 voxel per frame, no voxel ever
accumulated weight 8 and 93 % of the decode degenerated to the masked constant.  Here the SAME analytic
surface is a static scene and the camera pans over it.

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


YAW_STEP_DEG = 0.5      # camera yaw change per frame
YAW_AMPLITUDE_DEG = 4.0  # the camera pans back and forth inside +-4 degrees


def yaw_deg(t):
    """Triangle-wave pan: 0 -> +4 -> -4 -> ... in 0.5 degree steps (the scene stays inside the
    volume and is re-observed, so voxel weights accumulate as in a real scan)."""
    period = int(round(4 * YAW_AMPLITUDE_DEG / YAW_STEP_DEG))
    k = t % period
    q = period // 4
    if k <= q:
        return YAW_STEP_DEG * k
    if k <= 3 * q:
        return YAW_STEP_DEG * (2 * q - k)
    return YAW_STEP_DEG * (k - 4 * q)


def pose(t):
    """T_wc(t) = translate(0, 0, -1.5) . R_y(yaw(t))."""
    a = math.radians(yaw_deg(t))
    T = np.eye(4)
    T[:3, :3] = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    T[:3, 3] = [0.0, 0.0, -1.5]
    return T


def scene_depth0(u, v):
    """The static scene, given as the depth map camera 0 sees (defined for all real pixel
    coordinates of the 640x480 reference camera): 1.5 + 0.2 sin(u/40) cos(v/30) metres."""
    return 1.5 + 0.2 * np.sin(u / 40.0) * np.cos(v / 30.0)


_CLEAN = {}


def _clean_depth(t, H, W):
    """Noise-free depth (float64 metres) of the static scene from pose(t).  It depends on the yaw only, and the pan
    visits 17 yaw values, so it is computed once per (yaw, size) -- a 390-frame multi-GPU bench run builds its
    frames in seconds instead of a minute."""
    key = (yaw_deg(t), H, W)
    if key not in _CLEAN:
        sx, sy = 640.0 / W, 480.0 / H
        fx, fy, cx, cy = INTRINSICS[0, 0], INTRINSICS[1, 1], INTRINSICS[0, 2], INTRINSICS[1, 2]
        v, u = np.meshgrid(np.arange(H, dtype=np.float64) * sy, np.arange(W, dtype=np.float64) * sx, indexing="ij")
        R = pose(t)[:3, :3]
        ray = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=0).reshape(3, -1)
        r = (R @ ray).reshape(3, H, W)
        u0 = fx * r[0] / r[2] + cx
        v0 = fy * r[1] / r[2] + cy
        _CLEAN[key] = scene_depth0(u0, v0) / r[2]
    return _CLEAN[key]


def depth_image(t, H=480, W=640, seed=0):
    """Depth image of the static scene from pose(t), + N(0, 0.002) m sensor noise, stored as
    uint16 millimetres like the datasets (common.py:93).  All cameras share one optical centre, so
    the view from camera t is an exact homography of camera 0's: ray r = R_t K^-1 [u, v, 1] meets
    the scene at camera-0 pixel (u', v') = K r / r_z, and its depth in camera t is d0(u', v') / r_z."""
    rng = np.random.default_rng(seed + 1000 * t)
    d = _clean_depth(t, H, W) + rng.normal(0.0, 0.002, size=(H, W))
    return np.round(d * 1000.0).astype(np.uint16).astype(np.float64) / 1000.0


def depth_to_input_pts(depth, intr, T_wc, max_depth=10.0):
    """Host (numpy float64) version of the front end; same operation order as csrc/frontend.hip."""
    depth = np.asarray(depth, dtype=np.float64)
    mask = (depth > 0) & (depth < max_depth)
    depth = depth * mask
    H, W = depth.shape
    fx, fy, cx, cy = intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2]
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    xyz = np.stack([(u - cx) / fx * depth, (v - cy) / fy * depth, depth], axis=0)
    p = np.pad(xyz, ((0, 0), (1, 1), (1, 1)), mode="edge")
    gx = (((((p[:, :-2, 2:] + 2 * p[:, 1:-1, 2:]) + p[:, 2:, 2:]) - p[:, :-2, :-2]) - 2 * p[:, 1:-1, :-2])
          - p[:, 2:, :-2]) / 8.0
    gy = (((((p[:, 2:, :-2] + 2 * p[:, 2:, 1:-1]) + p[:, 2:, 2:]) - p[:, :-2, :-2]) - 2 * p[:, :-2, 1:-1])
          - p[:, :-2, 2:]) / 8.0
    n = np.stack([gx[1] * gy[2] - gx[2] * gy[1], gx[2] * gy[0] - gx[0] * gy[2], gx[0] * gy[1] - gx[1] * gy[0]])
    norm = np.sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2])
    n = n / np.maximum(norm, 1e-12)
    ur = ((np.arange(W, dtype=np.float32) - np.float32(cx)) / np.float32(fx)).astype(np.float64)
    vr = ((np.arange(H, dtype=np.float32) - np.float32(cy)) / np.float32(fy)).astype(np.float64)
    pc = np.stack([ur[None, :] * depth, vr[:, None] * depth, depth], axis=0)
    T = np.asarray(T_wc, dtype=np.float64)
    pw = [((T[i, 0] * pc[0] + T[i, 1] * pc[1]) + T[i, 2] * pc[2]) + T[i, 3] for i in range(3)]
    nw = [(T[i, 0] * n[0] + T[i, 1] * n[1]) + T[i, 2] * n[2] for i in range(3)]
    return np.stack(pw + nw, axis=-1).reshape(-1, 6)[mask.reshape(-1)]


def depth_u16(t, H=480, W=640, seed=0):
    """The frame's depth image as the dataset stores it: uint16 millimetres."""
    return np.round(depth_image(t, H, W, seed) * 1000.0).astype(np.uint16)


def intrinsics(H=480, W=640):
    intr = INTRINSICS.copy()
    if (H, W) != (480, 640):
        intr[0] *= W / 640.0
        intr[1] *= H / 480.0
    return intr


def frame(t, H=480, W=640, seed=0):
    """-> numpy float32 [1, N, 6] (the float64 -> .float() cast of run_e2e.py:249)."""
    intr = INTRINSICS.copy()
    if (H, W) != (480, 640):
        intr[0] *= W / 640.0
        intr[1] *= H / 480.0
    pts = depth_to_input_pts(depth_image(t, H, W, seed), intr, pose(t))
    return pts.astype(np.float32)[None]
