"""bnv_fusion_amd -- MI355X (gfx950) implementation of BNV-Fusion's per-frame local-geometry fusion
and SDF decode hot path, behind the reference's LitFusionPointNet / SparseVolume call surface.
See DESIGN.md (layout, kernels, rooflines) and INTEGRATION.md (how run_e2e.py picks it up)."""
import os as _os


def configure_runtime(hw_queues=8):
    """Optional, and only effective BEFORE the process's first HIP call (e.g. before the first ``torch.cuda`` use): asks
    the HIP runtime for ``hw_queues`` hardware queues (environment variable GPU_MAX_HW_QUEUES, unless it is set
    already).  The runtime serves a process's streams from 4 queues by default, and streams that share a queue run
    strictly in submission order; the frame pipelines here use up to four streams of their own next to the caller's
    (and RCCL's), so with 4 queues some of them would serialise.  ``streams.concurrent_stream`` verifies every side
    stream it hands out either way and says so (``.bnv_concurrent``) -- results never depend on this, only the
    overlap.  Importing the package does NOT touch the environment; bench.py, the tools and the tests call this first.
    Returns True if the setting will take effect."""
    import torch
    if torch.cuda.is_initialized():
        return _os.environ.get("GPU_MAX_HW_QUEUES") == str(hw_queues)
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(hw_queues)))
    return _os.environ["GPU_MAX_HW_QUEUES"] == str(int(hw_queues))


from .fusion import LitFusionPointNet, LocalNeRFModel, get_neighbors, load_pretrained  # noqa: F401
from .sparse_volume import SparseVolume, VolumeList, get_world_range  # noqa: F401
from .neural_map import NeuralMap  # noqa: F401
from . import optimize  # noqa: F401
from ._lib import BnvError  # noqa: F401


MLP_MODE_FP32_EXACT = 0     # v_mfma_f32_32x32x2_f32
MLP_MODE_SPLIT_F16 = 1      # fp32 operands split into f16 hi + lo, 3 products on the f16 MFMA (default)
MLP_MODE_TCNN = 2           # tiny-cuda-nn fp16 networks (selected automatically by tiny_cuda models)
MLP_MODE_F16 = 3            # fp32 checkpoint with operands rounded to f16: 1 product, 3x fewer MFMAs, SDF error ~1e-5


def set_mlp_mode(mode):
    """The package DEFAULT arithmetic of the MLP kernels for fp32-checkpoint models that do not name one themselves
    (include/bnv_fusion.h: bnv_set_mlp_mode; ``model.set_mlp_mode(m)`` pins one model).  Every call carries its mode in
    its bnv_grid_t, so models of different arithmetic -- tiny-cuda-nn ones always run mode 2 -- and host threads do not
    interfere; only models that follow this default change with it."""
    from . import _lib
    _lib.check(_lib.load().bnv_set_mlp_mode(int(mode)), "bnv_set_mlp_mode")
    if int(mode) in (0, 1, 3):
        _lib.fp32_mode = int(mode)


def get_mlp_mode():
    from . import _lib
    return int(_lib.load().bnv_get_mlp_mode())
