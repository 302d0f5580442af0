"""Spatial-hash sharded mode on ONE GPU, as rank 0 of a simulated world of W ranks: what a frame costs a rank of a
W-GPU node in that mode, and how many host waits it takes.

    python tools/spatial_single_rank.py [--world 8] [--grid 256] [--frames 200] [--in-flight 2]

Rank 0 of W voxelises the whole frame (replicated), encodes + upserts only the 1/W of the voxels it owns (the upsert
launch appends its boundary records), runs the frame's ONE all-gather (a real RCCL call on a one-rank group; the other
ranks' blocks are simulated by W - 1 copies of its own block, which the install kernel processes like foreign ones),
installs and decodes the voxels it owns -- through the product path (HipShardBackend over the C frame pipeline).
Reported: wall clock per frame with `--in-flight` frames enqueued ahead (the pipelined figure a node would run at if
every rank keeps this pace), the same with one frame at a time (latency), the host's enqueue time per frame, the
MLP kernels' own durations (HIP events) and the rank's share of the frame's work.  The volume is pre-rolled like the
bench (30 frames)."""
import argparse, ctypes as C, os, socket, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bnv_fusion_amd as bnv
from bnv_fusion_amd import synthetic, _lib
from bnv_fusion_amd import distributed as D

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--in-flight", type=int, default=2)
ap.add_argument("--checkpoint", default="fp32", choices=["fp32", "tcnn"])
ap.add_argument("--trace", action="store_true", help="HIP-event timeline of the two streams over a few frames")
ap.add_argument("--reserve", type=int, default=0, help="CUs the persistent MLP kernels leave to other streams")
args = ap.parse_args()
W = args.world
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dims, voxel = synthetic.GRID_DIMS[args.grid]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel, tiny_cuda=args.checkpoint == "tcnn")
POOL = 64
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)}
          for t in range(30 + POOL)]
be = D.HipShardBackend(np.array([dims] * 3), voxel, model, 0, W, capacity=1 << 21, device="cuda:0", tsdf=True, n_slots=max(4, args.in_flight + 2))
be.inputs_resident = True
be.copy_results = False
torch.cuda.synchronize()
lib = _lib.load()
if args.reserve:
    _lib.check(lib.bnv_set_option(b"reserve_cus", args.reserve), "reserve_cus")
stats = {"waits": 0, "recv": 0, "own": 0, "evals": 0, "enq": 0.0, "n": 0}


HOST = {k: 0.0 for k in ("begin", "bound", "upsert", "all_gather", "simulate", "finish")}
ranks_i32 = torch.arange(W, dtype=torch.int32, device="cuda:0")


TRACE = []


def ev(stream):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream)
    return e


def enqueue(fr, decode=True):
    if args.trace and be.pipe is not None:
        E, M = be.pipe.enc, be.pipe.main
        e0 = ev(E); f = be.encode(fr); e1 = ev(E)
        bound = be.bound(f)
        cap = -(-bound // D.REC_QUANTUM) * D.REC_QUANTUM
        m0 = ev(M); send = be.upsert(f, cap, decode); m1 = ev(M)
        one = be.recv_buffer(W * send.numel())
        dist.all_gather_into_tensor(one[: send.numel()], send)
        blocks = one.view(W, cap + 1, D.REC_WORDS)
        blocks[1:] = blocks[0]
        blocks[:, 0, 1] = ranks_i32
        be.install(f, one, cap)
        m2 = ev(M)
        h = be.finish(f, be.decode(f) if decode else None, 0)
        m3 = ev(M)
        TRACE.append((e0, e1, m0, m1, m2, m3))
        return h
    t0 = time.perf_counter()
    f = be.encode(fr)
    t1 = time.perf_counter()
    bound = be.bound(f); stats["waits"] += 1                     # the frame's one host wait
    t2 = time.perf_counter()
    cap = -(-bound // D.REC_QUANTUM) * D.REC_QUANTUM
    send = be.upsert(f, cap, decode)
    t3 = t4 = t5 = time.perf_counter()
    if cap:
        one = be.recv_buffer(W * send.numel())
        dist.all_gather_into_tensor(one[: send.numel()], send)   # the collective call itself (1-rank group)
        t4 = time.perf_counter()
        blocks = one.view(W, cap + 1, D.REC_WORDS)
        blocks[1:] = blocks[0]                                   # the other ranks' blocks: copies, sender ids patched
        blocks[:, 0, 1] = ranks_i32
        be.install(f, one, cap)
        stats["recv"] += one.numel() * 4
        t5 = time.perf_counter()
    h = be.finish(f, be.decode(f) if decode else None, 0)
    t6 = time.perf_counter()
    for k, d in zip(HOST, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
        HOST[k] += d
    stats["enq"] += t6 - t0 - (t5 - t4)                          # (without the simulation of the other ranks)
    return h


def collect(h):
    c, s = be.result(h)
    stats["own"] += 0 if c is None else len(c)
    stats["evals"] += be._last_evals
    stats["n"] += 1


def run(idx, in_flight, decode=True):
    pend = []
    for t in idx:
        while len(pend) >= in_flight:
            collect(pend.pop(0))
        pend.append(enqueue(frames[t], decode))
    while pend:
        collect(pend.pop(0))


with torch.no_grad():
    run(range(30), 2, decode=False)
    run(range(30, 38), 2)
    idx = [30 + (i % POOL) for i in range(args.frames)]
    for k in stats:
        stats[k] = 0
    for k in HOST:
        HOST[k] = 0.0
    lib.bnv_profile_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(idx, args.in_flight)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, cnt = (C.c_double * 4)(), (C.c_int64 * 4)()
    lib.bnv_profile_read(ms, cnt)
    lib.bnv_profile_enable(0)
    n = stats["n"]
    print(f"rank 0 of a simulated world of {W}, {args.grid}^3, 640x480, {args.checkpoint} networks, {n} frames, "
          f"{args.in_flight} in flight, {args.reserve} CUs reserved:")
    print(f"  pipelined wall clock  {1e3 * dt / n:.3f} ms per frame  -> {n / dt:.0f} frames/s for the rank set if every "
          f"rank keeps this pace")
    print(f"  host enqueue time     {1e3 * stats['enq'] / n:.3f} ms per frame (includes the bound wait); host waits per "
          f"frame: {stats['waits'] / n:.2f}")
    print("  host time per frame by phase (ms): " + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in HOST.items())
          + "  ('simulate' = this tool's stand-in for the other ranks' blocks, not part of a real rank's frame)")
    print(f"  MLP kernels (HIP events, overlapping streams): point encoder {ms[0] / max(cnt[0], 1):.3f} ms, "
          f"lattice table {ms[1] / max(cnt[1], 1):.3f} ms")
    print(f"  voxels owned per frame {stats['own'] / n:.0f}; SDF-MLP evaluations {stats['evals'] / n:.0f}; bytes received "
          f"per frame {stats['recv'] / n / 1e6:.2f} MB ({W} blocks)")
    if args.trace:
        TRACE.clear()
        run(idx[:12], args.in_flight)
        torch.cuda.synchronize()
        base = TRACE[4][0]
        print("  stream timeline (us from frame 4's encode start): E = encode stream [begin .. end], M = main stream "
              "[upsert start, upsert end, exchange end, finish end]")
        for k, (e0, e1, m0, m1, m2, m3) in enumerate(TRACE[4:10]):
            t = [1e3 * base.elapsed_time(x) for x in (e0, e1, m0, m1, m2, m3)]
            print(f"    frame {k + 4}: E [{t[0]:7.1f} .. {t[1]:7.1f}]   M [{t[2]:7.1f}, {t[3]:7.1f}, {t[4]:7.1f}, {t[5]:7.1f}]")
        args.trace = False
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(idx[:60], 1)
    torch.cuda.synchronize(); dt1 = time.perf_counter() - t0
    print(f"  one frame at a time   {1e3 * dt1 / 60:.3f} ms per frame (latency of a frame incl. the host round trips)")
dist.destroy_process_group()
