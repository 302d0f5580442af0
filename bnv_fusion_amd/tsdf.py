"""TSDFVolume -- drop-in for the reference's third_parties/fusion.py:19-300 on MI355X
(SURVEY.md section 8 f-1).  Same constructor and ``integrate`` / ``get_volume`` surface; the volumes live on
the GPU (``get_volume`` copies to host arrays like the reference's GPU mode does)."""
import ctypes as C

import numpy as np
import torch

from . import _lib


class TSDFVolume:
    def __init__(self, vol_bnds, voxel_size, use_gpu=True, device="cuda:0"):
        vol_bnds = np.asarray(vol_bnds, dtype=np.float64).copy()
        assert vol_bnds.shape == (3, 2), "[!] `vol_bnds` should be of shape (3, 2)."
        self._dev = torch.device(device)
        self._lib = _lib.require_device(self._dev.index or 0)
        self._vol_bnds = vol_bnds
        self._voxel_size = float(voxel_size)
        self._trunc_margin = 5 * self._voxel_size                      # fusion.py:36
        self._color_const = 256 * 256
        self._vol_dim = np.ceil((vol_bnds[:, 1] - vol_bnds[:, 0]) / self._voxel_size).copy(order="C").astype(int)
        self._vol_bnds[:, 1] = self._vol_bnds[:, 0] + self._vol_dim * self._voxel_size
        self._vol_origin = self._vol_bnds[:, 0].copy(order="C").astype(np.float32)
        dims = tuple(int(v) for v in self._vol_dim)
        # fusion.py:51-55: tsdf initialised to -trunc_margin (ones * 0 - margin), weights and colour to 0
        self.tsdf = torch.full(dims, -self._trunc_margin, dtype=torch.float32, device=self._dev)
        self.weight = torch.zeros(dims, dtype=torch.float32, device=self._dev)
        self.color = torch.zeros(dims, dtype=torch.float32, device=self._dev)
        self.gpu_mode = True

    def integrate(self, color_im, depth_im, cam_intr, cam_pose, obs_weight=1., max_depth=None, gate=None):
        """fusion.py:208-250.  depth_im [H, W] metres (numpy or tensor); color_im [H, W, 3] in [0, 255] or None.
        ``max_depth``: samples at or beyond it are invalid -- the reference's loader has already zeroed them when
        the frame reaches this call (common.py:110-113); here the raw image is passed and masked in the kernel.
        ``gate``: device int32 tensor; the launch does nothing when it holds 0 (pipelined NeuralMap: the frame's
        in-bounds point count, which the reference tests on the host before this call, run_e2e.py:91-92)."""
        depth = torch.as_tensor(depth_im)
        u16 = depth.dtype in (torch.uint16, torch.int16)     # the dataset's millimetres: converted in the kernel
        depth = depth.to(self._dev).contiguous() if u16 else depth.to(self._dev, torch.float32).contiguous()
        im_h, im_w = int(depth.shape[0]), int(depth.shape[1])
        col = self._fold_color(color_im)
        dim = (C.c_int32 * 3)(*[int(v) for v in self._vol_dim])
        org = (C.c_float * 3)(*self._vol_origin.tolist())
        intr = (C.c_float * 9)(*np.asarray(cam_intr, dtype=np.float64)[:3, :3].reshape(-1).astype(np.float32).tolist())
        pose = (C.c_float * 16)(*np.asarray(cam_pose, dtype=np.float64).reshape(-1).astype(np.float32).tolist())
        fn = self._lib.bnv_tsdf_integrate_u16 if u16 else self._lib.bnv_tsdf_integrate
        _lib.check(fn(
            _lib.ptr(self.tsdf), _lib.ptr(self.weight), _lib.ptr(self.color if col is not None else None), dim, org,
            np.float32(self._voxel_size), np.float32(self._trunc_margin), _lib.ptr(depth), _lib.ptr(col), im_h, im_w,
            intr, pose, float(obs_weight), float(max_depth or 0.0), _lib.ptr(gate), _lib.stream_ptr()), "bnv_tsdf_integrate")

    def _fold_color(self, color_im):
        """[H, W, 3] colour in [0, 255] -> the folded b*65536 + g*256 + r image of fusion.py:223-224 (or None)."""
        if color_im is None:
            return None
        c = torch.as_tensor(color_im).to(self._dev, torch.float32)
        return torch.floor(c[..., 2] * self._color_const + c[..., 1] * 256 + c[..., 0]).contiguous()

    BATCH_MAX = 8     # BNV_TSDF_BATCH_MAX

    def integrate_batch(self, depth_ims, cam_intrs, cam_poses, obs_weight=1., max_depth=None, color_ims=None):
        """``integrate`` for several consecutive uint16-millimetre depth frames of one size (``color_ims``: their colour
        images or None, per frame), one launch per BATCH_MAX frames; results identical to one call per frame in order."""
        ims = [torch.as_tensor(d) for d in depth_ims]
        if not ims:
            return
        cols = list(color_ims) if color_ims is not None else [None] * len(ims)
        if any(d.dtype not in (torch.uint16, torch.int16) or d.shape != ims[0].shape for d in ims):
            for d, k, p, c in zip(ims, cam_intrs, cam_poses, cols):
                self.integrate(c, d, k, p, obs_weight, max_depth)
            return
        cols = [self._fold_color(c) for c in cols]
        have_col = any(c is not None for c in cols)
        ims = [d.to(self._dev).contiguous() for d in ims]
        im_h, im_w = int(ims[0].shape[0]), int(ims[0].shape[1])
        dim = (C.c_int32 * 3)(*[int(v) for v in self._vol_dim])
        org = (C.c_float * 3)(*self._vol_origin.tolist())
        for g0 in range(0, len(ims), self.BATCH_MAX):
            grp = ims[g0: g0 + self.BATCH_MAX]
            k = len(grp)
            intr = np.stack([np.asarray(c, dtype=np.float64)[:3, :3].reshape(-1).astype(np.float32)
                             for c in cam_intrs[g0: g0 + k]]).reshape(-1)
            pose = np.stack([np.asarray(p, dtype=np.float64).reshape(-1).astype(np.float32)
                             for p in cam_poses[g0: g0 + k]]).reshape(-1)
            cg = cols[g0: g0 + k]
            _lib.check(self._lib.bnv_tsdf_integrate_batch_u16(
                _lib.ptr(self.tsdf), _lib.ptr(self.weight), _lib.ptr(self.color if have_col else None), dim, org,
                np.float32(self._voxel_size), np.float32(self._trunc_margin), k,
                (C.c_void_p * k)(*[d.data_ptr() for d in grp]),
                (C.c_void_p * k)(*[(c.data_ptr() if c is not None else 0) for c in cg]) if have_col else None,
                im_h, im_w,
                (C.c_float * (9 * k))(*intr.tolist()), (C.c_float * (16 * k))(*pose.tolist()), float(obs_weight),
                float(max_depth or 0.0), _lib.stream_ptr()), "bnv_tsdf_integrate_batch_u16")

    def get_volume(self):
        return self.tsdf.cpu().numpy(), self.color.cpu().numpy()

    def sdf_delta(self, truncated_dist, sdf_delta_weight=1.0):
        """NeuralMap.prepare_tsdf_volume (run_e2e.py:169-186) without leaving the GPU:
        tsdf * (voxel * 5), clipped to +-truncated_dist, times sdf_delta_weight -> [1, 1, X, Y, Z]."""
        v = self.tsdf * (self._voxel_size * 5)
        v = torch.clip(v[None, None], min=-truncated_dist, max=truncated_dist)
        return v * sdf_delta_weight
