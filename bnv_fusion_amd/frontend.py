"""Depth image -> ``input_pts`` on the GPU (the producer side of the hot path).

Replaces the float64 numpy/kornia code of FusionInferenceAbstractDataset.__getitem__
(src/datasets/fusion_inference_dataset.py:40-90) that the reference runs on DataLoader workers and
then uploads (7.4 MB/frame): here the 0.6 MB uint16 depth image is what crosses PCIe.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

DEPTH_DTYPES = {torch.uint16: 0, torch.int16: 0, torch.float32: 1, torch.float64: 2}


def depth_to_input_pts(depth, intr_mat, T_wc, max_depth=10.0, compact=True):
    """depth [H, W] on the GPU: uint16/int16 millimetres (the dataset PNGs), or float32/float64 metres.
    intr_mat 3x3, T_wc 4x4 (host, float64).  Returns input_pts [1, N, 6] float32 (valid pixels in
    row-major order) -- what ``frame['input_pts'].cuda().float()`` is in run_e2e.py:247-249.
    ``compact=False`` skips the host read of N and returns ([1, H*W, 6], n_valid device tensor) with the
    rows past n_valid filled with NaN (encode_pointcloud's bounds mask drops them)."""
    if not depth.is_cuda:
        raise _lib.BnvError("depth_to_input_pts runs on the GPU only")
    lib = _lib.require_device(depth.device.index or 0)
    d = depth.contiguous()
    H, W = int(d.shape[-2]), int(d.shape[-1])
    dt = DEPTH_DTYPES[d.dtype]
    K = (C.c_double * 9)(*np.asarray(intr_mat, dtype=np.float64)[:3, :3].reshape(-1))
    T = (C.c_double * 16)(*np.asarray(T_wc, dtype=np.float64).reshape(-1))
    ws = torch.empty(int(lib.bnv_depth_workspace_bytes(H, W)), dtype=torch.uint8, device=d.device)
    out = torch.empty((H * W, 6), dtype=torch.float32, device=d.device)
    n = torch.empty(1, dtype=torch.int32, device=d.device)           # written by the scan kernel
    fn = lib.bnv_depth_to_points if compact else lib.bnv_depth_to_points_padded   # padded: NaN rows behind n
    _lib.check(fn(_lib.ptr(d), dt, H, W, K, T, float(max_depth), _lib.ptr(ws), ws.numel(),
                                       _lib.ptr(out), _lib.ptr(n), _lib.stream_ptr()), "bnv_depth_to_points")
    if not compact:
        return out.unsqueeze(0), n
    return out[: int(n.item())].unsqueeze(0)
