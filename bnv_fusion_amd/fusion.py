"""LitFusionPointNet -- drop-in for the reference class of the same name
(src/models/fusion/local_point_fusion.py:21-65, 81-165, 265-379, 647-673) on MI355X.

Same call surface as run_e2e.py uses: construction from the Hydra-style cfg,
``load_state_dict`` with the reference's checkpoint keys, ``eval()/cuda()/freeze()``,
``encode_pointcloud``, ``_integrate``, ``decode_feature_grid_w_pts``, ``decode_implicit`` and a
``.nerf`` with ``xyz_encoding / geo_forward / get_neighbors``.  The arithmetic runs in the HIP
kernels of csrc/; this file only marshals torch tensors.  Both checkpoints are supported: the fp32 networks
(``model.tiny_cuda=False``, pointnet.ckpt) and the tiny-cuda-nn fp16 networks of the reference's default
configuration (``tiny_cuda=True``, pointnet_tcnn.ckpt).  There is no CPU fallback.
"""
import ctypes as C
import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, weights
from .sparse_volume import make_grid

_CORNER_IS_CEIL = ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1))


def encode_error_message(code):
    """bnv_encode_counters_t.error -> text."""
    return "bnv_encode_pointcloud: " + {
        1: "more touched voxels than the workspace was sized for",
        2: "output capacity exceeded",
        3: "a normal beyond the range certified for the f16-split point encoder (or NaN): encode in exact fp32 -- "
           "bnv_fusion_amd.set_mlp_mode(0)"}.get(int(code), f"error {int(code)}")


def _get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def get_neighbors(points, as_int=False):
    """[b, n, s, 3] -> [b, 8, n, s, 3] floor/ceil corners in the reference order
    (modules.py:586-655 with ``.int()``; fusion/utils.py:98-167 without)."""
    fl, ce = torch.floor(points), torch.ceil(points)
    out = torch.stack([torch.stack([(ce if cx else fl)[..., 0], (ce if cy else fl)[..., 1],
                                    (ce if cz else fl)[..., 2]], dim=-1) for cx, cy, cz in _CORNER_IS_CEIL], dim=1)
    return out.int() if as_int else out


class LocalNeRFModel(nn.Module):
    """The SDF decoder's parameters + the small helper surface SparseVolume.decode_pts and the
    global optimiser call (modules.py:81-123, 586-662, 923-971).  ``sdf_pack`` is the pre-permuted
    operand buffer the HIP decode kernels read."""

    def __init__(self, feat_dims=8, hidden_size=256, num_layers=4, num_encoding_fn_xyz=1, **_):
        super().__init__()
        if (feat_dims, hidden_size, num_layers, num_encoding_fn_xyz) != (8, 256, 4, 1):
            raise NotImplementedError("HIP decode kernels are built for 17->256x4->1 (fusion_pointnet_model.yaml)")
        dims = [3 + 6 * num_encoding_fn_xyz + feat_dims] + [hidden_size] * num_layers
        for i in range(num_layers):
            setattr(self, f"geo_layer{i}", nn.Linear(dims[i], dims[i + 1]))
        self.fc_alpha = nn.Linear(hidden_size, 1)
        self.mlp_mode = None     # arithmetic of the decode kernels: None = the package default (LitFusionPointNet.set_mlp_mode)
        self.num_layers = num_layers
        self.register_buffer("sdf_pack", torch.zeros(int(_lib.load().bnv_sdfmlp_pack_floats())), persistent=False)
        # transposed layers for the backward of decode_pts w.r.t. the volume features (global optimiser)
        self.register_buffer("sdf_bwd_pack", torch.zeros(int(_lib.load().bnv_sdfmlp_bwd_pack_floats())),
                             persistent=False)

    def repack(self):
        sd = {"nerf." + k: v for k, v in self.state_dict().items()}
        self.sdf_pack.copy_(torch.from_numpy(weights.pack_sdf_mlp(sd)))
        self.sdf_bwd_pack.copy_(torch.from_numpy(weights.pack_sdf_mlp_bwd(sd)))
        # sdf_pack is rewritten IN PLACE: whoever caches values computed from it (the frame pipe's persistent lattice
        # tables) compares this counter, not the buffer's address
        self.pack_version = getattr(self, "pack_version", 0) + 1

    @staticmethod
    def xyz_encoding(t):
        return torch.cat([t, torch.sin(t * 1.0), torch.cos(t * 1.0)], dim=-1)

    def get_neighbors(self, points):
        return get_neighbors(points, as_int=True)

    def geo_forward(self, xyz):
        """modules.py:657-662 on an arbitrary [..., 17] tensor (torch ops; differentiable).  The
        per-frame decode does not come through here -- see SparseVolume.decode_pts."""
        for i in range(self.num_layers):
            xyz = F.relu(getattr(self, f"geo_layer{i}")(xyz))
        return self.fc_alpha(xyz)


class _TcnnParams(nn.Module):
    """Holds ``model.params`` like tinycudann.NetworkWithInputEncoding does (checkpoint key ``*.model.params``)."""

    def __init__(self, n):
        super().__init__()
        self.params = nn.Parameter(torch.zeros(n))


class TcnnNeRFModel(nn.Module):
    """SDF decoder of the reference's default checkpoint (tcnnNeRFModel, modules.py:136-253): identity
    encoding padded to 32 with 1.0, FullyFusedMLP 64 x 3 -> 16, fp16.  ``sdf_pack`` feeds MLP mode 2."""
    mlp_mode = 2

    def __init__(self):
        super().__init__()
        self.model = _TcnnParams(64 * 32 + 64 * 64 * 2 + 16 * 64)
        self.register_buffer("sdf_pack", torch.zeros(12288 // 2), persistent=False)
        self.register_buffer("sdf_bwd_pack", torch.zeros(int(_lib.load().bnv_sdfmlp_tcnn_bwd_pack_floats())),
                             persistent=False)   # transposed layers (backward)

    def repack(self):
        p = self.model.params.detach().cpu().numpy()
        self.sdf_pack.copy_(torch.from_numpy(weights.pack_sdf_tcnn(p)))
        self.sdf_bwd_pack.copy_(torch.from_numpy(weights.pack_sdf_tcnn_bwd(p)))
        self.pack_version = getattr(self, "pack_version", 0) + 1      # (as LocalNeRFModel.repack)

    xyz_encoding = staticmethod(LocalNeRFModel.xyz_encoding)

    def get_neighbors(self, points):
        return get_neighbors(points, as_int=True)

    def geo_forward(self, xyz):
        """modules.py:249-253 (torch emulation of the fp16 network; the per-frame decode runs in HIP)."""
        shapes = list(xyz.shape)
        x = xyz.reshape(-1, shapes[-1]).half()
        x = torch.cat([x, torch.ones((x.shape[0], 32 - x.shape[1]), dtype=x.dtype, device=x.device)], 1)
        off = 0
        dims = [32, 64, 64, 64, 16]
        for i in range(4):
            w = self.model.params[off: off + dims[i + 1] * dims[i]].reshape(dims[i + 1], dims[i]).half()
            off += dims[i + 1] * dims[i]
            x = (x.float() @ w.float().t())
            x = (torch.relu(x) if i < 3 else x).half()
        return x[:, :1].reshape(shapes[:-1] + [1])


class LitFusionPointNet(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.cfg = cfg
        model = _get(cfg, "model")
        trainer = _get(cfg, "trainer")
        self.tiny_cuda = bool(_get(model, "tiny_cuda", False))
        self.dense_volume = bool(_get(trainer, "dense_volume", False)) if trainer is not None else False
        self.feat_dims = int(_get(model, "feature_vector_size", 8))
        nerf_cfg = _get(model, "nerf") or {}
        nerf_kwargs = dict(nerf_cfg) if isinstance(nerf_cfg, dict) else {k: getattr(nerf_cfg, k) for k in (
            "hidden_size", "num_layers", "num_encoding_fn_xyz") if hasattr(nerf_cfg, k)}
        self.interpolate_decode = bool(_get(nerf_cfg, "interpolate_decode", True))
        if self.tiny_cuda:
            # the networks are fixed by src/models/tcnn_config.json (FullyFusedMLP, 64 neurons, 3 hidden layers)
            self.nerf = TcnnNeRFModel()
            self.pointnet_backbone = nn.Module()
            self.pointnet_backbone.model = _TcnnParams(64 * 16 + 64 * 64 * 2 + 16 * 64)
        else:
            self.nerf = LocalNeRFModel(self.feat_dims, **{k: nerf_kwargs[k] for k in (
                "hidden_size", "num_layers", "num_encoding_fn_xyz") if k in nerf_kwargs})
            self.pointnet_backbone = _PointNetParams(self.feat_dims)
        self.voxel_size = _get(model, "voxel_size")
        self.min_pts_in_grid = int(_get(model, "min_pts_in_grid", 8))
        self.training_global = bool(_get(model, "training_global", False))
        self.shard = (0, 1, 3)   # (rank, world, block_log2) of the spatial sharding; see distributed.py
        self.register_buffer("pointnet_pack", torch.zeros(int(_lib.load().bnv_pointnet_pack_floats())),
                             persistent=False)
        self._enc_ws = None
        self._enc_ws_key = None
        self._enc_ws_points = 0
        self._grid_cache = {}
        self._mlp_mode = None    # fp32 checkpoints: None = the package default (set_mlp_mode pins this model)

    # ---- nn.Module plumbing --------------------------------------------------------------------
    @property
    def device(self):
        return self.pointnet_pack.device

    def load_state_dict(self, state_dict, strict=True):
        sd = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in state_dict.items()
              if not k.startswith("nerf.color_layer") and not k.startswith("nerf.fc_rgb")}
        res = super().load_state_dict(sd, strict=strict)
        self.repack()
        return res

    def repack(self):
        self.pointnet_pack.zero_()
        if self.tiny_cuda:
            w = torch.from_numpy(weights.pack_pointnet_tcnn(self.pointnet_backbone.model.params.detach().cpu().numpy()))
        else:
            w = torch.from_numpy(weights.pack_pointnet({k: v for k, v in self.state_dict().items()}))
        self.pointnet_pack[: w.numel()].copy_(w)
        self.nerf.repack()

    @property
    def mlp_mode(self):
        """Arithmetic of this model's MLP kernels (include/bnv_fusion.h: bnv_set_mlp_mode): 2 for tiny-cuda-nn
        checkpoints; else the mode ``set_mlp_mode`` gave this model, else None = the package default
        (bnv_fusion_amd.set_mlp_mode).  It is passed with every call (bnv_grid_t.mlp_mode), never through a global."""
        return 2 if self.tiny_cuda else self._mlp_mode

    def set_mlp_mode(self, mode):
        """Pins THIS model (encoder and decoder) to an fp32-checkpoint arithmetic mode: 0 exact fp32, 1 split f16,
        3 f16 operands; None: follow the package default again."""
        if self.tiny_cuda:
            raise _lib.BnvError("a tiny-cuda-nn checkpoint runs in MLP mode 2 only")
        if mode is not None and int(mode) not in (0, 1, 3):
            raise ValueError("mlp mode of an fp32 checkpoint: 0, 1 or 3")
        self._mlp_mode = None if mode is None else int(mode)
        self.nerf.mlp_mode = self._mlp_mode
        return self

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()
        return self

    def _lib_for(self, t):
        if not t.is_cuda:
            raise _lib.BnvError("bnv_fusion_amd runs on the GPU only: move the model and its inputs to cuda "
                                "(there is no CPU fallback)")
        return _lib.require_device(t.device.index or 0)

    def _grid(self, n_xyz, bound_min, bound_max, voxel_size):
        """bnv_grid_t for these arguments.  Cached while the caller keeps passing the SAME tensor
        objects (the volume's own attributes, unmodified), so the per-frame call costs no
        device->host reads; anything else is re-read."""
        args = (n_xyz, bound_min, bound_max)
        c = self._grid_cache
        mode = _lib.model_mode(self)
        if c and c["voxel"] == float(voxel_size) and c["extra"] == (self.min_pts_in_grid, self.shard, mode) and all(
                isinstance(a, torch.Tensor) and r() is a and a._version == v
                for a, r, v in zip(args, c["refs"], c["versions"])):
            return c["grid"], c["res"]
        res = [int(v) for v in n_xyz]
        grid = make_grid(res, bound_min, bound_max, voxel_size, self.min_pts_in_grid, self.shard, mlp_mode=mode)
        if all(isinstance(a, torch.Tensor) for a in args):
            self._grid_cache = {"voxel": float(voxel_size), "extra": (self.min_pts_in_grid, self.shard, mode),
                                "refs": [weakref.ref(a) for a in args], "versions": [a._version for a in args],
                                "grid": grid, "res": res}
        return grid, res

    # ---- encode (local_point_fusion.py:81-165) ----------------------------------------------------
    def _encode_buffers(self, lib, dev, n, res, emit_all, out):
        """Workspace (kept across frames, zero-filled = clean) and output buffers of one encode of up to n points."""
        nvox = res[0] * res[1] * res[2]
        n_arr = (C.c_int32 * 3)(*res)
        key = (tuple(res), dev)
        if self._enc_ws is None or self._enc_ws_key != key or self._enc_ws_points < n:
            self._enc_ws_points = max(n, 1024) if self._enc_ws_key != key else max(n, 2 * self._enc_ws_points)
            need = int(lib.bnv_encode_workspace_bytes(self._enc_ws_points, n_arr))
            self._enc_ws = torch.zeros(need, dtype=torch.uint8, device=dev)   # zero-filled = clean
            self._enc_ws_key = key
        if out is not None:
            feats, pcounts, flat_ids, grid_ids = out
            cap = int(grid_ids.shape[0])
            assert feats.shape == (cap, 8) and feats.dtype == torch.float32 and feats.is_contiguous()
            assert grid_ids.shape == (cap, 3) and grid_ids.dtype == torch.int64 and grid_ids.is_contiguous()
            assert pcounts.dtype == torch.int64 and pcounts.numel() >= cap and pcounts.is_contiguous()
            assert flat_ids.dtype == torch.int64 and flat_ids.numel() >= cap and flat_ids.is_contiguous()
        else:
            cap = min(8 * n, nvox) if emit_all else min(8 * n // max(self.min_pts_in_grid, 1) + 1, nvox)
            cap = max(cap, 1)
            feats = torch.empty((cap, 8), dtype=torch.float32, device=dev)
            pcounts = torch.empty(cap, dtype=torch.int64, device=dev)
            flat_ids = torch.empty(cap, dtype=torch.int64, device=dev)
            grid_ids = torch.empty((cap, 3), dtype=torch.int64, device=dev)
        counters = torch.empty(8, dtype=torch.int32, device=dev)    # written in full by the encode's last kernel
        return feats, pcounts, flat_ids, grid_ids, counters, cap

    def shard_boundary_counts(self):
        """Device int32 [shard_world] view into the encode workspace: touched boundary voxels owned by each rank,
        valid between the two halves of a sharded encode (include/bnv_fusion.h: bnv_encode_begin)."""
        off = int(_lib.load().bnv_encode_shard_counts_offset())
        return self._enc_ws[off: off + 4 * self.shard[1]].view(torch.int32)

    def encode_pointcloud_async(self, input_pts, n_xyz, bound_min, bound_max, voxel_size, emit_all=False, out=None,
                                between=None):
        """Enqueues the encode and returns WITHOUT synchronising: capacity-sized output buffers
        (feats [cap,8], pcounts [cap] i64, flat_ids [cap] i64, grid_ids [cap,3] i64) and the device
        counters (int32 [8]: n_valid, n_unique, n_out, n_avg_pts as float bits, error).  Downstream kernels
        read n_out from ``counters[2:3]`` on the device.  ``out``: caller-provided contiguous buffers
        (feats, pcounts, flat_ids, grid_ids) to write into; their row count is the capacity.  ``between``: called
        between the two halves of the encode (voxelise + sorted-unique | PointNet + reduction), e.g. to enqueue a
        read-back of shard_boundary_counts()."""
        lib = self._lib_for(self.pointnet_pack)
        assert input_pts.dim() == 3 and input_pts.shape[0] == 1 and input_pts.shape[2] == 6
        pts = input_pts[0].detach().float().contiguous()
        n = int(pts.shape[0])
        grid, res = self._grid(n_xyz, bound_min, bound_max, voxel_size)
        feats, pcounts, flat_ids, grid_ids, counters, cap = self._encode_buffers(lib, pts.device, n, res, emit_all, out)
        ws = (_lib.ptr(self._enc_ws), self._enc_ws.numel(), self._enc_ws_points)
        _lib.check(lib.bnv_encode_begin(_lib.ptr(pts), n, C.byref(grid), *ws, _lib.stream_ptr()), "bnv_encode_begin")
        if between is not None:
            between()
        _lib.check(lib.bnv_encode_finish(_lib.ptr(pts), n, C.byref(grid), _lib.ptr(self.pointnet_pack), *ws,
                                         _lib.ptr(feats), _lib.ptr(pcounts), _lib.ptr(flat_ids), _lib.ptr(grid_ids),
                                         cap, 1 if emit_all else 0, _lib.ptr(counters), _lib.stream_ptr()),
                   "bnv_encode_finish")
        return feats, pcounts, flat_ids, grid_ids, counters, cap

    def encode_depth_async(self, depth, intr_mat, T_wc, max_depth, n_xyz, bound_min, bound_max, voxel_size, out=None,
                           between=None):
        """encode_pointcloud_async straight from a depth image [H, W] on the GPU (uint16/int16 millimetres, or
        float32/float64 metres): the front end (FusionInferenceAbstractDataset.__getitem__,
        fusion_inference_dataset.py:40-90) is fused in front of the voxelisation -- one kernel computes every pixel's
        world point + normal in float64, writes its float32 input_pts row (pixel order, NaN rows for invalid pixels)
        and marks its voxels.  Returns the same tuple as encode_pointcloud_async plus the input_pts tensor
        [1, H*W, 6]."""
        from .frontend import DEPTH_DTYPES
        lib = self._lib_for(self.pointnet_pack)
        if not depth.is_cuda:
            raise _lib.BnvError("encode_depth_async runs on the GPU only")
        d = depth.contiguous()
        H, W = int(d.shape[-2]), int(d.shape[-1])
        n = H * W
        grid, res = self._grid(n_xyz, bound_min, bound_max, voxel_size)
        feats, pcounts, flat_ids, grid_ids, counters, cap = self._encode_buffers(lib, d.device, n, res, False, out)
        pts = torch.empty((n, 6), dtype=torch.float32, device=d.device)
        K = (C.c_double * 9)(*np.asarray(intr_mat, dtype=np.float64)[:3, :3].reshape(-1))
        T = (C.c_double * 16)(*np.asarray(T_wc, dtype=np.float64).reshape(-1))
        ws = (_lib.ptr(self._enc_ws), self._enc_ws.numel(), self._enc_ws_points)
        _lib.check(lib.bnv_encode_begin_depth(_lib.ptr(d), DEPTH_DTYPES[d.dtype], H, W, K, T, float(max_depth),
                                              C.byref(grid), *ws, _lib.ptr(pts), _lib.stream_ptr()),
                   "bnv_encode_begin_depth")
        if between is not None:
            between()
        _lib.check(lib.bnv_encode_finish_image(_lib.ptr(pts), n, W, C.byref(grid), _lib.ptr(self.pointnet_pack), *ws,
                                               _lib.ptr(feats), _lib.ptr(pcounts), _lib.ptr(flat_ids),
                                               _lib.ptr(grid_ids), cap, 0, _lib.ptr(counters), _lib.stream_ptr()),
                   "bnv_encode_finish_image")
        return feats, pcounts, flat_ids, grid_ids, counters, cap, pts.unsqueeze(0)

    def encode_pointcloud(self, input_pts, n_xyz, bound_min, bound_max, voxel_size, return_dense=True):
        lib = self._lib_for(self.pointnet_pack)
        feats, pcounts, flat_ids, grid_ids, counters, cap = self.encode_pointcloud_async(
            input_pts, n_xyz, bound_min, bound_max, voxel_size, emit_all=return_dense)
        pts = input_pts[0].detach().float().contiguous()
        dev = pts.device
        n = int(pts.shape[0])
        grid, res = self._grid(n_xyz, bound_min, bound_max, voxel_size)
        host = counters.cpu()                       # the one device->host sync of the frame
        n_valid, n_unique, n_out, err = int(host[0]), int(host[1]), int(host[2]), int(host[4])
        if err:
            raise _lib.BnvError(encode_error_message(err))
        if n_valid == 0:                            # local_point_fusion.py:101-102
            return None, None, None, None, None
        if not return_dense:
            n_avg_pts = counters[3:4].view(torch.float32)[0]
            return feats[:n_out], pcounts[:n_out].unsqueeze(-1), flat_ids[:n_out], grid_ids[:n_out], n_avg_pts
        # dense contract (local_point_fusion.py:127-141): grids + unique ids + per-pair flat ids
        g = grid_ids[:n_out]
        feat_grids = torch.zeros((1, 8, *res), dtype=torch.float32, device=dev)
        mask = torch.zeros((1, 1, *res), dtype=torch.float32, device=dev)
        mask[0, 0, g[:, 0], g[:, 1], g[:, 2]] = pcounts[:n_out].float()
        feat_grids[0, :, g[:, 0], g[:, 1], g[:, 2]] = feats[:n_out].t()
        bmask = torch.empty(n, dtype=torch.uint8, device=dev)
        _lib.check(lib.bnv_voxelize_pairs(_lib.ptr(pts), n, C.byref(grid), None, None, None, _lib.ptr(bmask),
                                          _lib.stream_ptr()), "bnv_voxelize_pairs")
        kept = pts[bmask.bool()].contiguous()
        pair_ids = torch.empty(8 * kept.shape[0], dtype=torch.int64, device=dev)
        _lib.check(lib.bnv_voxelize_pairs(_lib.ptr(kept), int(kept.shape[0]), C.byref(grid), None,
                                          _lib.ptr(pair_ids), None, None, _lib.stream_ptr()), "bnv_voxelize_pairs")
        return feat_grids, mask, flat_ids[:n_out], pair_ids.unsqueeze(0)

    def get_relative_xyz(self, xyz, bound_min, voxel_size, n_xyz=(1, 1, 1)):
        """local_point_fusion.py:153-165: xyz [1, N, 3] -> (relative_xyz [1, 8, N, 3] f32,
        grid_id [1, 8, N, 3] i32)."""
        lib = self._lib_for(xyz)
        n = int(xyz.shape[1])
        pts = torch.zeros((n, 6), dtype=torch.float32, device=xyz.device)
        pts[:, :3] = xyz[0]
        grid = make_grid(n_xyz, bound_min, bound_min, voxel_size, self.min_pts_in_grid)
        gid = torch.empty((8 * n, 3), dtype=torch.int32, device=xyz.device)
        rel = torch.empty((8 * n, 3), dtype=torch.float32, device=xyz.device)
        _lib.check(lib.bnv_voxelize_pairs(_lib.ptr(pts), n, C.byref(grid), _lib.ptr(gid), None, _lib.ptr(rel), None,
                                          _lib.stream_ptr()), "bnv_voxelize_pairs")
        return rel.reshape(1, 8, n, 3), gid.reshape(1, 8, n, 3)

    # ---- integrate (local_point_fusion.py:647-673) ------------------------------------------------
    def _integrate(self, volume_object, fine_coords, fine_feats, fine_weights):
        volume_object.integrate(fine_coords, fine_feats, fine_weights)

    # ---- dense decode (local_point_fusion.py:265-379) ---------------------------------------------
    def decode_feature_grid_w_pts(self, voxel_coords, feat_grid, pts_weight, voxel_size, bound_min,
                                  gradient=False, global_coords=True):
        """All three branches of the reference (local_point_fusion.py:265-367), one launch each: global_coords=True
        (the signature default: trilinear features, one evaluation at coords / (res - 1), unscaled); else
        interpolate_decode=True (8 corner evaluations, the yaml configuration) or False (one evaluation at the nearest
        voxel).  Returns (sdf [1, Q], neighbor_feats) like the reference: [1, 8, Q, F] for the corner branch,
        [1, Q, F] for the other two."""
        if gradient:
            # the reference's own branch (:367-369) reads the undefined name `mask` and raises NameError
            raise NotImplementedError("decode_feature_grid_w_pts(gradient=True): the reference's branch cannot run "
                                      "(local_point_fusion.py:368 uses an undefined name); use SparseVolume.decode_pts "
                                      "for gradients")
        lib = self._lib_for(voxel_coords)
        q = voxel_coords.detach().reshape(-1, 3).float().contiguous()
        n = int(q.shape[0])
        fg = feat_grid.detach().float().contiguous()
        pw = pts_weight.detach().float().contiguous()
        dims = (C.c_int32 * 3)(*[int(v) for v in fg.shape[-3:]])
        out = torch.empty(n, dtype=torch.float32, device=q.device)
        variant = 2 if global_coords else (0 if self.interpolate_decode else 1)
        nf1 = torch.empty((n, 8), dtype=torch.float32, device=q.device) if variant else None
        status = torch.zeros(2, dtype=torch.int32, device=q.device)
        _lib.check(lib.bnv_decode_dense_mode(_lib.ptr(fg), _lib.ptr(pw), dims, float(np.float32(voxel_size)),
                                             self.min_pts_in_grid, _lib.ptr(self.nerf.sdf_pack), _lib.ptr(q), n, variant,
                                             _lib.model_mode(self) + 1, _lib.ptr(out),
                                             _lib.ptr(nf1) if variant else None, _lib.ptr(status),
                                             _lib.stream_ptr()), "bnv_decode_dense_mode")
        if int(status[1]) == 5:
            raise RuntimeError("decode_feature_grid_w_pts: a feature is outside the range the split-f16 MLP arithmetic "
                               "is certified for (or not finite); call bnv_set_mlp_mode(0) (exact fp32) for this grid")
        if variant:
            return out.reshape(1, n), nf1.reshape(1, n, 8)
        # the reference also returns the gathered neighbour features [1, 8, Q, F]
        nb = get_neighbors(voxel_coords.unsqueeze(1), as_int=True).squeeze(2).long()
        X, Y, Z = [int(v) for v in fg.shape[-3:]]
        inside = ((nb >= 0) & (nb < torch.tensor([X, Y, Z], device=nb.device))).all(-1)
        nbc = nb.clamp(min=0)
        nbc = torch.minimum(nbc, torch.tensor([X - 1, Y - 1, Z - 1], device=nb.device))
        nf = fg[0][:, nbc[..., 0], nbc[..., 1], nbc[..., 2]].permute(1, 2, 3, 0) * inside.unsqueeze(-1)
        return out.reshape(1, n), nf

    def decode_implicit(self, feat_grid, points, normalize, voxel_size=None, mask=None, test=False):
        """local_point_fusion.py:372-379 via LocalNeRFModel.forward (modules.py:941-971); torch ops."""
        if normalize:
            points = points / voxel_size
        enc = self.nerf.xyz_encoding(points[..., :3])
        if not test:
            feat_grid = feat_grid.unsqueeze(1).repeat(1, points.shape[1], 1)
        geo_in = torch.cat([enc, feat_grid], dim=-1)
        if mask is not None:
            flat = geo_in.reshape(-1, geo_in.shape[-1])
            m = mask.reshape(-1)
            out = torch.zeros_like(flat[:, :1])
            out[m] = self.nerf.geo_forward(flat[m])
            pred = out.reshape(list(geo_in.shape[:-1]) + [1])
        else:
            pred = self.nerf.geo_forward(geo_in)
        return pred * voxel_size if normalize else pred


class _PointNetParams(nn.Module):
    """Parameter container with the reference's PointNetEncoder names (pointnet_utils.py:230-244)."""

    def __init__(self, feat_dims):
        super().__init__()
        self.conv1 = nn.Conv1d(6, 128, 1)
        self.conv2 = nn.Conv1d(128, 128, 1)
        self.conv3 = nn.Conv1d(128, 128, 1)
        self.conv4 = nn.Conv1d(128, feat_dims, 1)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm1d(128), nn.BatchNorm1d(128), nn.BatchNorm1d(128)
        self.bn4 = nn.BatchNorm1d(feat_dims)


def load_pretrained(device="cuda:0", voxel_size=0.01, min_pts_in_grid=8, path=None, tiny_cuda=False):
    """Model with a converted reference checkpoint (weights/pointnet_fp32.npz, or pointnet_tcnn.npz when
    ``tiny_cuda``), frozen, on device."""
    cfg = {"trainer": {"dense_volume": False},
           "model": {"feature_vector_size": 8, "voxel_size": voxel_size, "tiny_cuda": tiny_cuda,
                     "min_pts_in_grid": min_pts_in_grid,
                     "nerf": {"hidden_size": 256, "num_layers": 4, "num_encoding_fn_xyz": 1,
                              "interpolate_decode": True}}}
    model = LitFusionPointNet(cfg)
    model.load_state_dict(weights.load_npz(path or (weights.DEFAULT_TCNN if tiny_cuda else weights.DEFAULT_FP32)))
    model.eval()
    model.to(device)
    model.freeze()
    return model
