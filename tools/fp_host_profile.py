import socket, sys, time, collections
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, distributed as D
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(90)]
fp = D.FrameParallelNeuralMap(np.array([dims]*3), voxel, model, device="cuda:0", tsdf=True)
for h in fp.process_stream([[f] for f in frames[:30]], decode=False): pass
fp.flush(); torch.cuda.synchronize()
T = collections.defaultdict(float)
def wrap(obj, name, key):
    orig = getattr(obj, name)
    def f(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); T[key] += time.perf_counter() - t0; return r
    setattr(obj, name, f)
be = fp.backend
R = 8
orig_v, orig_t = be.volume.integrate_batch, be.tsdf_vol.integrate_batch
def replayed(items):
    orig_v(items * R)
    extra = (R - 1) * sum(int(it[0].shape[0]) for it in items)
    be.volume._inflight -= extra; be.volume._rows_upper -= extra
be.volume.integrate_batch = replayed
be.tsdf_vol.integrate_batch = lambda d, k, p, obs_weight=1.: orig_t(d * R, k * R, p * R, obs_weight)
wrap(be, "integrate_records", "integrate_records x8 (host)")
wrap(be, "integrate_tsdf", "integrate_tsdf x8 (host)")
wrap(be, "encode_frame", "encode_frame (host)")
wrap(be, "pack", "pack (host)")
wrap(fp, "exchange", "exchange total (incl. the header wait)")
wrap(be, "decode_record", "decode_record (host)")
wrap(fp, "submit", "submit total")
wrap(fp, "finish", "finish total")
n = 60
t0 = time.perf_counter()
for last in fp.process_stream([[f] for f in frames[30:30+n]]): pass
t_enq = time.perf_counter() - t0
fp.flush(); last.result(); torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"host enqueue loop {1e3*t_enq/n:.3f} ms/batch; wall {1e3*tot/n:.3f} ms/batch")
for k, v in T.items(): print(f"  {k:28s} {1e3*v/n:.3f} ms/batch")
dist.destroy_process_group()
