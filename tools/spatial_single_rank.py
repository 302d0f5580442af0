"""Spatial-hash sharded mode (ShardedNeuralMap) as a ONE-rank RCCL group: the per-frame fixed costs of that mode
(collective calls, size reads, halo bookkeeping) on one GPU, against the plain NeuralMap frame."""
import socket, sys, time, collections
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
from bnv_fusion_amd import synthetic
from bnv_fusion_amd.distributed import ShardedNeuralMap
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(70)]
nm = ShardedNeuralMap(np.array([dims] * 3), voxel, model, device="cuda:0")
for f in frames[:30]:
    nm.integrate(f)
for f in frames[30:35]:
    nm.fuse_and_decode(f)
T = collections.defaultdict(float)
def wrap(obj, name):
    orig = getattr(obj, name)
    def f(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig(*a, **k); torch.cuda.synchronize(); T[name] += time.perf_counter() - t0; return r
    setattr(obj, name, f)
import bnv_fusion_amd.distributed as D
torch.cuda.synchronize(); t0 = time.perf_counter()
for f in frames[35:65]:
    nm.fuse_and_decode(f)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"spatial mode, 1 rank: {1e3 * dt / 30:.3f} ms per frame")
for n in ("encode_integrate", "tables_for", "install_and_blend"):
    wrap(nm.backend, n)
orig_ag = D.all_gather_var
def ag(t, group=None):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig_ag(t, group); torch.cuda.synchronize(); T["all_gather_var"] += time.perf_counter() - t0; return r
D.all_gather_var = ag
for f in frames[35:65]:
    nm.fuse_and_decode(f)
for k, v in T.items():
    print(f"  {k:20s} {1e3 * v / 30:.3f} ms per frame (synchronised)")
dist.destroy_process_group()
