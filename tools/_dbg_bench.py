import os, sys, time, runpy, collections
sys.path.insert(0, '.')
import torch
import bnv_fusion_amd as bnv
from bnv_fusion_amd import neural_map, sparse_volume, fusion, frontend, tsdf
T = collections.defaultdict(list)
def wrap(obj, name, key=None):
    orig = getattr(obj, name)
    key = key or name
    def f(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); T[key].append(time.perf_counter() - t0); return r
    setattr(obj, name, f)
wrap(neural_map, "frame_input_pts")
wrap(fusion.LitFusionPointNet, "encode_pointcloud_async")
wrap(sparse_volume.SparseVolume, "integrate")
wrap(sparse_volume.SparseVolume, "decode_lattice")
wrap(tsdf.TSDFVolume, "integrate", "tsdf_integrate")
_empty = torch.empty
def empty(*a, **k):
    t0 = time.perf_counter(); r = _empty(*a, **k); dt = time.perf_counter() - t0
    T["empty_pinned" if k.get("pin_memory") else "empty"].append(dt); return r
torch.empty = empty
sys.argv = ["bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-alt-mode"]
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    for k, v in T.items():
        big = [(i, round(1e3 * x, 2)) for i, x in enumerate(v) if x > 0.005]
        print(k, "calls", len(v), "slow (>5 ms):", big[:8], file=sys.stderr)
