"""Golden vectors for the TSDF side fusion, captured from the reference's own
CPU path.

Build-container only (needs /root/reference):  python tests/golden/make_golden_tsdf.py
third_parties/fusion.py falls back to a CPU implementation when pycuda is missing (FUSION_GPU_MODE = 0,
:225-291).  Its helpers are numba kernels; numba is not installed here, so ``njit`` is shimmed as the identity and
``prange`` as ``range`` -- the SAME Python source then runs un-jitted (numba's float32 arithmetic is IEEE like
numpy's).  Three 120x160 synthetic depth frames are integrated into a 1.0 m volume at 0.025 m (40^3 voxels).

The CPU path and the CUDA kernel the reference runs on a GPU box are two formulations of the same update: the CPU
path inverts the pose with numpy (float64) and rounds pixel coordinates half-to-even (np.round), the kernel applies
the transposed rotation in float32 and rounds half away from zero (roundf).  The HIP kernel follows the CUDA kernel;
tests compare against these goldens with a small tolerance and allow the handful of voxels whose projection lands
exactly between two pixels.

Only DATA is written (tests/golden/tsdf_40.npz).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_shims  # noqa: E402


def main():
    ref_shims.install()
    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: (a[0] if (a and callable(a[0]) and not k) else (lambda f: f))
    nb.prange = range
    sys.modules["numba"] = nb
    sys.path.insert(0, "/root/reference")
    from third_parties import fusion as ref_fusion
    assert ref_fusion.FUSION_GPU_MODE == 0
    from oracle import bnv_oracle as orc

    bounds = np.array([[-0.5, 0.5], [-0.5, 0.5], [-0.5, 0.5]])
    vol = ref_fusion.TSDFVolume(bounds.copy(), voxel_size=0.025, use_gpu=False)
    H, W = 120, 160
    intr = orc.SYNTHETIC_INTRINSICS.copy()
    intr[:2] *= 0.25
    depths, poses = [], []
    for t in (0, 3, 7):
        d = (orc.synthetic_depth(t, H, W) * 0.5).astype(np.float32)          # surface at ~0.75 m: inside the volume
        T = orc.synthetic_pose(t).copy()
        T[:3, 3] = [0.0, 0.0, -0.95]
        vol.integrate(np.zeros((H, W, 3)), d, intr, T, obs_weight=1.0)
        depths.append(d)
        poses.append(T)
    tsdf, _ = vol.get_volume()
    np.savez_compressed(os.path.join(HERE, "tsdf_40.npz"), bounds=bounds, voxel_size=0.025, intr=intr,
                        depths=np.stack(depths), poses=np.stack(poses), tsdf=tsdf.astype(np.float32),
                        weight=vol._weight_vol_cpu.astype(np.float32), origin=vol._vol_origin)
    print("tsdf_40:", tsdf.shape, "observed voxels", int((vol._weight_vol_cpu > 0).sum()),
          "weights up to", float(vol._weight_vol_cpu.max()))


if __name__ == "__main__":
    main()
