"""Frame-parallel multi-GPU mode measured as a ONE-rank RCCL group: what one rank does per batch (encode, header +
payload all-gather through RCCL, integrate x batch, decode), pipelined as in bench.py.  With `--replay N` the
rank also integrates N-1 extra copies of the record per batch -- the replicated work of an N-GPU run."""
import argparse, socket, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic
from bnv_fusion_amd.distributed import FrameParallelNeuralMap
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=60); ap.add_argument("--replay", type=int, default=1); ap.add_argument("--ahead", type=int, default=3); ap.add_argument("--reserve", type=int, default=0); ap.add_argument("--split", action="store_true"); args = ap.parse_args()
for _try in range(8):     # (a free port can be taken between the probe and the store's listen: try another)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        break
    except Exception as e:
        if "EADDRINUSE" not in str(e) or _try == 7:
            raise
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(30 + args.frames)]
fp = FrameParallelNeuralMap(np.array([dims]*3), voxel, model, device="cuda:0", tsdf=True)
fp.max_unsettled = args.ahead
from bnv_fusion_amd import _lib
if args.reserve:
    _lib.load().bnv_set_option(b"reserve_cus", args.reserve)
for h in fp.process_stream([[f] for f in frames[:30]], decode=False): pass
if args.replay > 1:     # emulate the replicated part of an N-rank batch: N - 1 more integrates (+ TSDF) per batch
    vol, tv = fp.backend.volume, fp.backend.tsdf_vol
    orig_v, orig_t = vol.integrate_batch, tv.integrate_batch
    if args.split:      # the rank's own frame sits in the middle of the batch: two batched upserts
        fp.rank_pos = args.replay // 2
    def replayed(items):
        k = len(items)
        if args.split:
            orig_v(items * (args.replay // 2)); orig_v(items * (args.replay - args.replay // 2))
        else:
            orig_v(items * args.replay)
        extra = (args.replay - 1) * sum(int(it[0].shape[0]) for it in items)
        vol._inflight -= extra; vol._rows_upper -= extra          # keep the host-side row bound consistent
    def replayed_t(d, k, p, obs_weight=1., max_depth=None, color_ims=None):
        orig_t(d * args.replay, k * args.replay, p * args.replay, obs_weight, max_depth,
               None if color_ims is None else list(color_ims) * args.replay)
    vol.integrate_batch, tv.integrate_batch = replayed, replayed_t
fp.flush(); torch.cuda.synchronize()
import ctypes as C
for rep in range(2):
    _lib.load().bnv_profile_enable(1)
    t0 = time.perf_counter(); last = None
    for last in fp.process_stream([[f] for f in frames[30:]]): pass
    fp.flush(); c, sdf = last.result(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pm, pn = (C.c_double * 4)(), (C.c_int64 * 4)(); _lib.load().bnv_profile_read(pm, pn); _lib.load().bnv_profile_enable(0)
    print(f"  kernels: pointnet {pm[0]/max(pn[0],1):.3f} ms, table MLP {pm[1]/max(pn[1],1):.3f} ms")
    print(f"ahead={args.ahead} one-rank frame-parallel over RCCL, {args.replay} integrates per batch: {1e3*dt/args.frames:.3f} ms per batch "
          f"-> an {args.replay}-rank run would do {args.replay*args.frames/dt:.0f} frames/s if the exchange hides")
dist.destroy_process_group()
