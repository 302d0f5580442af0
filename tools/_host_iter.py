import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
from bnv_fusion_amd import synthetic
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<20, device="cuda:0", tsdf=True)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(80)]
for f in frames[:30]: nm.fuse_and_decode_async(f, decode=False).result()
pending = None
for f in frames[30:35]:
    h = nm.fuse_and_decode_async(f)
    if pending is not None: pending.result()
    pending = h
pending.result(); torch.cuda.synchronize()
import os
from bnv_fusion_amd import _lib
if os.environ.get('PROF'): _lib.load().bnv_profile_enable(1)
ts = []; t0 = time.perf_counter(); pending = None
for f in frames[35:75]:
    a = time.perf_counter()
    h = nm.fuse_and_decode_async(f)
    b = time.perf_counter()
    if pending is not None: pending.result()
    c = time.perf_counter()
    ts.append((b - a, c - b))
    pending = h
pending.result(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{40/dt:.1f} fps; enqueue ms: " + " ".join(f"{1e3*x:.2f}" for x, _ in ts))
print("result-wait ms: " + " ".join(f"{1e3*y:.2f}" for _, y in ts))
