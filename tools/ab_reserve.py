import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, _lib
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<21, device="cuda:0", tsdf=True)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(80)]
for t in range(30): nm.integrate(frames[t])
lib = _lib.load()
def run(first, count):
    for t in range(first - 5, first): nm.fuse_and_decode(frames[t])
    torch.cuda.synchronize(); time.sleep(0.5); pending = None; t0 = time.perf_counter()
    for t in range(first, first + count):
        h = nm.fuse_and_decode_async(frames[t])
        if pending is not None: pending.result()
        pending = h
    pending.result(); torch.cuda.synchronize()
    return count / (time.perf_counter() - t0)
res = {}
for rnd in range(3):
    for r in (0, 4, 8, 16):
        lib.bnv_set_option(b"reserve_cus", r)
        res.setdefault(r, []).append(run(35, 40))
for r, v in res.items(): print("reserve", r, ["%.1f" % x for x in v])
