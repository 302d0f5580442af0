"""Import shims that let the upstream reference (/root/reference) run on CPU in
THIS container, so that golden vectors can be captured from it.

Test infrastructure only.  Nothing here is imported by the product package and
nothing here runs on the GPU box (the reference tree does not exist there).
The shims stand in for third-party wheels the reference imports but that are
absent from this image:

* behavioural: ``torch_scatter.scatter_mean``, ``pytorch_lightning.LightningModule``
  and a dict-backed ``open3d.core.HashMap`` / ``Tensor`` (only what
  ``SparseVolume`` calls: sparse_volume.py:531-594, 617-653, 682-691);
* inert: open3d, trimesh, skimage, tinycudann, cv2, kornia, imageio, omegaconf,
  hydra, commentjson.
"""
import os
import pickle
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


# --------------------------------------------------------------------------- #
# torch_scatter
# --------------------------------------------------------------------------- #
def _scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    index = index.expand_as(src) if index.shape != src.shape else index
    n = int(index.max()) + 1 if dim_size is None else dim_size
    shape = list(src.shape)
    shape[dim] = n
    total = torch.zeros(shape, dtype=src.dtype, device=src.device)
    total.scatter_add_(dim, index, src)
    count = torch.zeros(shape, dtype=src.dtype, device=src.device)
    count.scatter_add_(dim, index, torch.ones_like(src))
    return total / count.clamp(min=1)


# --------------------------------------------------------------------------- #
# open3d.core stand-in
# --------------------------------------------------------------------------- #
class _Dtype:
    Float32 = "f32"
    Int64 = "i64"
    Int32 = "i32"


class _O3Tensor:
    """Thin wrapper over a torch tensor with the handful of o3c.Tensor methods used."""

    def __init__(self, t):
        self.t = t

    @staticmethod
    def from_dlpack(t):
        if not isinstance(t, torch.Tensor):  # a DLPack PyCapsule from torch.utils.dlpack.to_dlpack
            t = torch.utils.dlpack.from_dlpack(t)
        return _O3Tensor(t)

    def to_dlpack(self):
        return self.t

    def to(self, dtype, *a, **k):
        if dtype in ("i64", torch.int64):
            return _O3Tensor(self.t.to(torch.int64))
        if dtype in ("i32", torch.int32):
            return _O3Tensor(self.t.to(torch.int32))
        if dtype in ("f32", torch.float32):
            return _O3Tensor(self.t.to(torch.float32))
        return self

    def cpu(self):
        return self

    def numpy(self):
        return self.t.detach().cpu().numpy()

    def __len__(self):
        return self.t.shape[0]

    def __eq__(self, other):
        return _O3Tensor(self.t == other)

    def __getitem__(self, idx):
        if isinstance(idx, _O3Tensor):
            idx = idx.t
        return _O3Tensor(self.t[idx])

    def __setitem__(self, idx, val):
        if isinstance(idx, _O3Tensor):
            idx = idx.t
        if isinstance(val, _O3Tensor):
            val = val.t
        self.t[idx] = val


class _HashMap:
    """dict-backed multi-value hash map with Open3D 0.14 insert/find semantics:
    ``insert`` does NOT overwrite existing keys and reports them in the mask."""

    def __init__(self, capacity, key_dtype=None, key_element_shape=None,
                 value_dtype=None, value_element_shape=None,
                 value_dtypes=None, value_element_shapes=None, device=None):
        if value_dtypes is None:
            value_dtypes = (value_dtype,)
            value_element_shapes = (value_element_shape,)
        self._kdim = int(key_element_shape[0])
        self._vdt = [torch.float32 if d == "f32" else torch.int64 for d in value_dtypes]
        self._vshape = [tuple(s) for s in value_element_shapes]
        self._cap = max(int(capacity), 1)
        self._alloc(self._cap)
        self._n = 0
        self._map = {}

    def _alloc(self, cap):
        self._keys = torch.zeros((cap, self._kdim), dtype=torch.int64)
        self._vals = [torch.zeros((cap,) + s, dtype=d) for s, d in zip(self._vshape, self._vdt)]

    def _grow(self, need):
        if need <= self._cap:
            return
        cap = max(need, 2 * self._cap)
        ok, ov = self._keys, self._vals
        self._alloc(cap)
        self._keys[: self._n] = ok[: self._n]
        for a, b in zip(self._vals, ov):
            a[: self._n] = b[: self._n]
        self._cap = cap

    def insert(self, keys, values):
        if not isinstance(values, (tuple, list)):
            values = (values,)
        k = keys.t.reshape(-1, self._kdim).to(torch.int64)
        m = k.shape[0]
        self._grow(self._n + m)
        buf = torch.zeros(m, dtype=torch.int32)
        mask = torch.zeros(m, dtype=torch.bool)
        kl = k.tolist()
        for i in range(m):
            key = tuple(kl[i])
            j = self._map.get(key)
            if j is None:
                j = self._n
                self._n += 1
                self._map[key] = j
                self._keys[j] = k[i]
                for a, v in zip(self._vals, values):
                    a[j] = v.t.reshape((m,) + a.shape[1:])[i]
                mask[i] = True
            buf[i] = j
        return _O3Tensor(buf), _O3Tensor(mask)

    def find(self, keys):
        k = keys.t.reshape(-1, self._kdim).to(torch.int64)
        m = k.shape[0]
        buf = torch.zeros(m, dtype=torch.int32)
        mask = torch.zeros(m, dtype=torch.bool)
        kl = k.tolist()
        for i in range(m):
            j = self._map.get(tuple(kl[i]))
            if j is not None:
                buf[i] = j
                mask[i] = True
        return _O3Tensor(buf), _O3Tensor(mask)

    def active_buf_indices(self):
        return _O3Tensor(torch.arange(self._n, dtype=torch.int32))

    def key_tensor(self):
        return _O3Tensor(self._keys)

    def value_tensor(self, i=0):
        return _O3Tensor(self._vals[i])


# --------------------------------------------------------------------------- #
# pytorch_lightning stand-in
# --------------------------------------------------------------------------- #
class _LightningModule(torch.nn.Module):
    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    def log(self, *a, **k):
        pass


class AttrDict(dict):
    """Attribute-access dict standing in for an omegaconf DictConfig."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __setattr__(self, k, v):
        self[k] = v


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    """Install the stubs into sys.modules and put the reference on sys.path."""
    if "/root/reference" not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    o3c = _mod("open3d.core", HashMap=_HashMap, Tensor=_O3Tensor, Dtype=_Dtype,
               Device=lambda d: d, int64="i64", int32="i32", float32="f32")
    _mod("open3d", core=o3c)
    for name in ("trimesh", "cv2", "imageio", "tinycudann", "kornia", "kornia.geometry",
                 "kornia.geometry.depth", "skimage", "skimage.transform"):
        _mod(name)
    _mod("skimage.measure", marching_cubes=lambda *a, **k: None,
         marching_cubes_lewiner=lambda *a, **k: None)
    _mod("torch_scatter", scatter_mean=_scatter_mean)
    _mod("omegaconf", DictConfig=dict, OmegaConf=object)
    _mod("hydra", main=lambda *a, **k: (lambda f: f))
    import json
    import re

    def _cj_load(fh):
        txt = re.sub(r"//.*", "", fh.read())
        txt = re.sub(r",\s*([}\]])", r"\1", txt)
        return json.loads(txt)

    _mod("commentjson", load=_cj_load)
    plu = _mod("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    plc = _mod("pytorch_lightning.callbacks")
    _mod("pytorch_lightning", LightningModule=_LightningModule, utilities=plu, callbacks=plc,
         seed_everything=lambda s, **k: torch.manual_seed(s))


def install_run_e2e():
    """Further inert stand-ins that importing the reference's CALLER (src/run_e2e.py and the dataset / helper modules
    it pulls in) needs on top of install(): names imported at module level from wheels absent here.  None of them is
    called on the recorded path (tests/golden/make_golden_caller.py)."""
    def absent(*a, **k):
        raise RuntimeError("stand-in for a wheel that is absent from this image")

    sys.modules["kornia.geometry.depth"].depth_to_normals = absent
    sys.modules["kornia.geometry.depth"].depth_to_3d = absent
    _mod("quaternion")
    _mod("plyfile", PlyData=object)
    _mod("numba", njit=lambda *a, **k: (lambda f: f), prange=range)


class _StubUnpickler(pickle.Unpickler):
    """Fabricates empty classes for non-torch modules pickled into the Lightning checkpoints."""

    def find_class(self, module, name):
        if module.split(".")[0] in ("torch", "collections", "numpy", "builtins", "_codecs"):
            return super().find_class(module, name)
        return type(name, (), {})


class _StubPickleModule:
    Unpickler = _StubUnpickler
    load = staticmethod(pickle.load)
    __name__ = "stub_pickle"


def load_checkpoint_state_dict(path):
    ckpt = torch.load(path, map_location="cpu", pickle_module=_StubPickleModule, weights_only=False)
    return ckpt["state_dict"]


def make_cfg(voxel_size, tiny_cuda=False):
    return AttrDict(
        device_type="cpu",
        trainer=dict(dense_volume=False),
        model=dict(
            feature_vector_size=8, voxel_size=voxel_size, tiny_cuda=tiny_cuda,
            training_global=False, global_coords=False,
            bound_min=[-1.0, -1.0, -1.0], bound_max=[1.0, 1.0, 1.0], min_pts_in_grid=8,
            point_net=dict(in_channels=6),
            nerf=dict(hidden_size=256, num_layers=4, num_encoding_fn_xyz=1, num_encoding_fn_dir=6,
                      include_input_xyz=True, include_input_dir=True, interpolate_decode=True,
                      global_coords=False, xyz_agnostic=False),
            loss=dict(bce_loss=1.0, reg_loss=0.001),
        ),
    )


def build_reference_model(voxel_size, workdir):
    """Returns (LitFusionPointNet in eval mode with pointnet.ckpt loaded, SparseVolume class)."""
    install()
    cwd = os.getcwd()
    os.makedirs(workdir, exist_ok=True)
    os.chdir(workdir)  # the ctor creates ./plots (local_point_fusion.py:47-49)
    try:
        from src.models.fusion.local_point_fusion import LitFusionPointNet
        from src.models.sparse_volume import SparseVolume
        model = LitFusionPointNet(make_cfg(voxel_size))
    finally:
        os.chdir(cwd)
    sd = load_checkpoint_state_dict(os.path.join(REFERENCE_ROOT, "pretrained", "pointnet.ckpt"))
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    model.eval()
    model.freeze()
    return model, SparseVolume
