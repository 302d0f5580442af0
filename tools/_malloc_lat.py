import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
from bnv_fusion_amd import synthetic
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<22, device="cuda:0", tsdf=True)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(60)]
for f in frames[:30]: nm.fuse_and_decode_async(f, decode=False).result()
hs = [nm.fuse_and_decode_async(f) for f in frames[30:34]]
for mb in (33, 35, 37, 8, 9):
    a0 = torch.cuda.memory_stats()["num_device_alloc"]
    t0 = time.perf_counter(); x = torch.empty(mb << 20, dtype=torch.uint8, device="cuda:0"); dt = time.perf_counter() - t0
    print(f"new {mb} MB block while the GPU is busy: {1e3*dt:.2f} ms (device allocs +{torch.cuda.memory_stats()['num_device_alloc'] - a0})")
for h in hs: h.result()
torch.cuda.synchronize()
for mb in (41, 43):
    t0 = time.perf_counter(); x = torch.empty(mb << 20, dtype=torch.uint8, device="cuda:0"); dt = time.perf_counter() - t0
    print(f"new {mb} MB block, GPU idle: {1e3*dt:.2f} ms")
