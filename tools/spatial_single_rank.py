"""Spatial-hash sharded mode on ONE GPU, as rank 0 of a simulated world of W ranks: what a frame costs a rank of an
W-GPU node in that mode, phase by phase, and how many host waits it takes.

    python tools/spatial_single_rank.py [--world 8] [--grid 256]

Rank 0 of W voxelises the whole frame (replicated), encodes + upserts only the 1/W of the voxels it owns, packs its
boundary records, runs the frame's ONE all-gather (a real RCCL call on a one-rank group; the other ranks' blocks are
simulated by W - 1 copies of its own block, which bnv_shard_install processes like foreign ones), installs, and
decodes the voxels it owns.  Phases are timed with HIP events on the stream (no host sync between phases); the
pipelined total is wall-clock over all frames.  The volume is pre-rolled like the bench (30 frames)."""
import argparse, os, socket, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bnv_fusion_amd as bnv
from bnv_fusion_amd import synthetic
from bnv_fusion_amd import distributed as D

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--frames", type=int, default=40)
args = ap.parse_args()
W = args.world
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
dims, voxel = synthetic.GRID_DIMS[args.grid]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)}
          for t in range(30 + args.frames)]
be = D.HipShardBackend(np.array([dims] * 3), voxel, model, 0, W, capacity=1 << 21, device="cuda:0", tsdf=True)
PH = ("encode", "upsert+pack", "all_gather", "(install: in finish)", "install+decode")
acc = {k: 0.0 for k in PH}
waits = recv_bytes = own = evals = 0


def frame(fr, decode=True, timed=False):
    global waits, recv_bytes, own, evals
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(PH) + 1)]
    ev[0].record()
    f = be.encode(fr)
    ev[1].record()
    bound = be.bound(f); waits += 1                      # the frame's one host wait
    cap = -(-bound // D.REC_QUANTUM) * D.REC_QUANTUM
    send = be.upsert(f, cap, decode)
    ev[2].record()
    one = torch.empty((1, send.numel()), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(one.view(-1), send)      # the collective call itself (1-rank group)
    recv = one.repeat(W, 1)                              # the other ranks' blocks: copies, sender ids patched
    recv.view(W, cap + 1, D.REC_WORDS)[:, 0, 1] = torch.arange(W, dtype=torch.int32, device=recv.device)
    ev[3].record()
    res = be.install(f, recv.view(-1), cap)
    ev[4].record()
    sdf = be.decode(f) if decode else None
    ev[5].record()
    h = be.finish(f, sdf, res)
    if timed:
        c, s = be.result(h)
        for i, k in enumerate(PH):
            acc[k] += ev[i].elapsed_time(ev[i + 1])
        recv_bytes += recv.numel() * 4
        own += 0 if c is None else len(c)
        evals += int(be.last_mlp_evals().item())
    return h


with torch.no_grad():
    for fr in frames[:30]:
        be.result(frame(fr, decode=False))
    for fr in frames[30:34]:
        be.result(frame(fr))
    n = 0
    for fr in frames[34:]:
        frame(fr, timed=True); n += 1
    print(f"rank 0 of a simulated world of {W}, {args.grid}^3, 640x480: per frame, phases in stream time (ms)")
    for k in PH:
        print(f"  {k:14s} {acc[k] / n:7.3f}")
    print(f"  sum            {sum(acc.values()) / n:7.3f}   host waits per frame: {waits / (30 + 4 + n):.2f}")
    print(f"  voxels owned per frame {own / n:.0f}; SDF-MLP evaluations {evals / n:.0f}; bytes received per frame "
          f"{recv_bytes / n / 1e6:.2f} MB ({W} blocks of {recv_bytes / n / W / 1e6:.3f} MB)")
    # pipelined: frame t+1 enqueued before frame t is collected
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pend = None
    for fr in frames[34:]:
        h = frame(fr)
        if pend is not None:
            be.result(pend)
        pend = h
    be.result(pend)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"  pipelined wall clock: {1e3 * dt / n:.3f} ms per frame  (-> {n / dt:.0f} frames/s per rank-set if every rank keeps this pace)")
dist.destroy_process_group()
