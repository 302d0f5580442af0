"""Host-side enqueue cost of the per-frame calls (no synchronisation inside the timed loops)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic
from bnv_fusion_amd.distributed import HipFrameBackend
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<20, device="cuda:0", tsdf=True)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(80)]
for t in range(30): nm.integrate(frames[t])
torch.cuda.synchronize()
t0 = time.perf_counter(); hs = [nm.fuse_and_decode_async(frames[t]) for t in range(30, 70)]; t1 = time.perf_counter()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"fuse_and_decode_async: host enqueue {1e3*(t1-t0)/40:.3f} ms/frame; GPU drained after {1e3*(t2-t0)/40:.3f} ms/frame")
for h in hs: h.result()
be = HipFrameBackend(np.array([dims]*3), voxel, model, device="cuda:0", tsdf=True)
from bnv_fusion_amd.distributed import header_counters
encs = [be.encode_frame(frames[t]) for t in range(8)]
torch.cuda.synchronize()
n_out = [int(header_counters(e.hdr.cpu())[2]) for e in encs]
rows = -(-max(n_out) // 1024) * 1024
pay = [be.pack(e, rows) for e in encs]
for name, fn in (("encode_frame", lambda i: be.encode_frame(frames[30 + i])),
                 ("pack", lambda i: be.pack(encs[i % 8], rows)),
                 ("integrate_record(+tsdf)", lambda i: be.integrate_record(encs[i % 8].hdr, pay[i % 8], rows, n_out[i % 8], frames[i % 8])),
                 ("decode_record", lambda i: be.decode_record(encs[i % 8].hdr, pay[i % 8], rows, n_out[i % 8]))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40): fn(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:26s} host {1e3*(t1-t0)/40:.3f} ms/call; with GPU drain {1e3*(t2-t0)/40:.3f} ms/call")
