"""bnv_fusion_amd -- MI355X (gfx950) implementation of BNV-Fusion's per-frame local-geometry fusion
and SDF decode hot path, behind the reference's LitFusionPointNet / SparseVolume call surface.
See DESIGN.md (layout, kernels, rooflines) and INTEGRATION.md (how run_e2e.py picks it up)."""
from .fusion import LitFusionPointNet, LocalNeRFModel, get_neighbors, load_pretrained  # noqa: F401
from .sparse_volume import SparseVolume, get_world_range  # noqa: F401
from .neural_map import NeuralMap  # noqa: F401
