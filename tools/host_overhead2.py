"""Host-side cost of enqueuing a frame with NeuralMap.fuse_and_decode_async: a burst of 4 frames into an empty
queue (no back-pressure from the GPU), repeated; and the split by call (cProfile, top entries)."""
import sys, time, numpy as np, torch, cProfile, pstats
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<20, device="cuda:0", tsdf=True)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(80)]
for t in range(30): nm.integrate(frames[t])
hs = [nm.fuse_and_decode_async(frames[t]) for t in range(30, 36)]
for h in hs: h.result()
tot = 0.0
for rep in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hs = [nm.fuse_and_decode_async(frames[36 + 4 * rep + i]) for i in range(4)]
    tot += time.perf_counter() - t0
    for h in hs: h.result()
print(f"host enqueue, bursts of 4 frames into an empty queue: {1e3 * tot / 32:.3f} ms per frame")
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
hs = [nm.fuse_and_decode_async(frames[70 + i]) for i in range(4)]
pr.disable()
for h in hs: h.result()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
