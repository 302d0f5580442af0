"""Spatially sharded mode on ONE GPU, as rank r of a world of W ranks: what a frame costs a rank of a W-GPU node in that
mode, with the REAL ghost rows of the other ranks.

    python tools/spatial_single_rank.py [--world 8] [--rank r | --all-ranks] [--grid 256] [--frames 200] [--in-flight 3]
                                        [--ownership region|first_touch|hash] [--block-log2 3] [--scene pan|sweep]

Two passes.  RECORD (`--record FILE`, a process of its own): all W shards of the volume live in this one process and run
the frames in lock step through the product path -- encode with ownership, bound, upsert (+ boundary records), the
exchange as a torch.stack of the W send blocks, install, decode -- exactly what W processes over gloo / RCCL do
(tests/test_gpu_multiprocess.py checks that equivalence); every frame's W send blocks are kept, and next to them what
every rank did (voxels, pairs, SDF-MLP evaluations) and what the single volume does for the same frames.  REPLAY
(`--ghosts FILE`): rank r alone, timed: it voxelises the whole frame (replicated), encodes + upserts the voxels it owns,
runs the frame's ONE all-gather (a real RCCL call on a one-rank group that lands its own block) into a buffer that
already holds the OTHER ranks' recorded blocks for that frame (one device copy: the stand-in for the collective's
payload), installs the adjacent records as ghost rows and decodes the voxels it owns.  Round 4's tool filled the other
ranks' blocks with copies of the rank's own block: no foreign voxel ever became a ghost row, lattice points with a
foreign corner stayed dead and a rank of 8 was priced at 171 k evaluations where a real one does ~270 k.

`--all-ranks`: the record pass, then every rank in a process of its own, then the MAX over the ranks (a rank set runs
at the pace of its slowest member).  Reported per rank: wall clock per frame with `--in-flight` frames enqueued ahead,
the host's enqueue time, the MLP kernels' own durations (HIP events), the rank's share of the work -- and the
evaluation count of the record pass beside the replay's (they must agree: same frames, same ghosts).  `--frames` >= 2000
gives the sustained figure (the package heats up over the first ~1000 frames)."""
import argparse, ctypes as C, gc, os, socket, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, _lib
from bnv_fusion_amd import distributed as D

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=0, help="the rank of the simulated world this GPU plays")
ap.add_argument("--all-ranks", action="store_true", help="every rank of the world in turn (fresh shard each), then the max")
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--in-flight", type=int, default=3)
ap.add_argument("--checkpoint", default="fp32", choices=["fp32", "tcnn"])
ap.add_argument("--trace", action="store_true", help="HIP-event timeline of the two streams over a few frames")
ap.add_argument("--reserve", type=int, default=0, help="CUs the persistent MLP kernels leave to other streams")
ap.add_argument("--ownership", default=None, help="ownership rule of the shards (default: the package's)")
ap.add_argument("--block-log2", type=int, default=None, help="block edge of the sharding, log2 voxels (default: the package's)")
ap.add_argument("--axis", type=int, default=None, help="region rule: axis the first bands are stacked along")
ap.add_argument("--scene", default="pan", choices=["pan", "sweep"], help="the bench's panning camera or the room sweep (sequence.py)")
ap.add_argument("--preroll", type=int, default=None, help="frames fused (not decoded) before the pool of 64 timed frames "
                "(default: 30 for the pan, like the bench; 230 for the sweep: the camera has turned a third of the way by then "
                "and the region rule has met the sweep)")
ap.add_argument("--record", default=None, help="(internal) record pass: all W shards in this process, blocks saved to this file")
ap.add_argument("--ghosts", default=None, help="recorded blocks of the other ranks (from --record); without it --rank records first")
ap.add_argument("--exchange", default="torch", choices=["torch", "on_stream"],
                help="torch: the collective call of torch.distributed (a one-rank group here: ProcessGroupNCCL runs it on its "
                     "own stream, two event hops); on_stream: what a collective enqueued on the main stream itself amounts to "
                     "for one rank (a device copy of the own block, no hop)")
ap.add_argument("--encoder-wgs", type=int, default=None, help="workgroups of the persistent point encoder (default: the package's)")
ap.add_argument("--timeline", type=int, default=0, help="GPU timestamps of every stage (bnv_frame_timeline) over this many "
                "pipelined frames after the timed run")
ap.add_argument("--no-latency", action="store_true")
ap.add_argument("--exchange-delay", type=float, default=0.0,
                help="microseconds a single-wave spin kernel adds behind the stand-in all-gather on the stream it runs on "
                     "(the latency of a real 8-rank collective; bnv_probe_spin at 2.1 GHz)")
ap.add_argument("--ahead", type=int, default=1, help="1: the next frame's encode is enqueued before the host waits for "
                "this frame's exchange bound (ShardedNeuralMap's next_frame); 0: the loop of round 3")
ap.add_argument("--json", action="store_true", help="(internal) print the rank's figures as one JSON line at the end")
args = ap.parse_args()
W = args.world

if args.all_ranks:
    # every rank in a process of its own, one after the other -- as on a node, where a rank IS a process with one frame
    # pipeline (several pipelines created one after the other in ONE process end up, now and then, with streams that
    # share a hardware queue: 0.46 instead of 0.27 ms per frame on a random rank; a fresh process never showed it).  The
    # parent never touches the GPU.
    import json, subprocess, tempfile
    rows = []
    base = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--all-ranks"]
    ghosts = args.ghosts or os.path.join(tempfile.gettempdir(), f"bnv_ghosts_{os.getpid()}.pt")
    if not args.ghosts:
        # the record pass: all W shards in one process, real exchange, the blocks of every frame saved
        out = subprocess.run(base + ["--record", ghosts], capture_output=True, text=True, timeout=1800)
        print(out.stdout, end="")
        if out.returncode != 0:
            print(out.stderr[-3000:])
            raise SystemExit("record pass failed")
        base += ["--ghosts", ghosts]
    # a short throw-away run first: the first process on a fresh box pays for cold file caches and clocks
    subprocess.run(base + ["--rank", "0", "--frames", "200", "--no-latency"], capture_output=True, text=True, timeout=900)
    for r in range(W):
        cmd = base + ["--rank", str(r), "--json", "--no-latency"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
        lines = out.stdout.splitlines()
        js = [l for l in lines if l.startswith("{\"rank\"")]
        if out.returncode != 0 or not js:
            print(out.stdout[-2000:], out.stderr[-2000:])
            raise SystemExit(f"rank {r} failed")
        print("\n".join(l for l in lines if not l.startswith("{\"rank\"")))
        rows.append(json.loads(js[-1]))
    print(f"\nall {W} ranks ({args.frames} frames each, {args.in_flight} in flight; one process per rank, one after the other):")
    print("  rank   ms/frame   (median, slowest segment)   voxels owned   pairs encoded   MLP evaluations (record pass)   encoder ms   table ms   host ms")
    for o in rows:
        print(f"  {o['rank']:4d}   {o['ms']:8.3f}   ({o['ms_med']:.3f}, {o['ms_worst']:.3f})            {o['own']:12.0f}   "
              f"{o['pairs']:13.0f}   {o['evals']:15.0f} ({o.get('evals_rec', 0):9.0f})   {o['enc_ms']:10.3f}   {o['tab_ms']:8.3f}   {o['host_ms']:7.3f}")
    if rows and rows[0].get("evals_rec"):
        print("  (the timed replay cycles the pool frames for a long time: its live set, and with it a rank's evaluations, "
              "sits above the record pass's, whose ratios above are the clean measure of duplicated work; with the "
              "persistent tables a rank then evaluates ~4 % fewer entries than it reads)")
    for k, name in (("own", "voxels owned"), ("pairs", "pairs encoded"), ("evals", "MLP evaluations")):
        v = np.array([o[k] for o in rows])
        print(f"  {name}: max / mean over the ranks {v.max() / max(v.mean(), 1e-9):.3f}")
    v = np.array([o["ms"] for o in rows])
    print(f"  ms per frame: mean {v.mean():.3f}, MAX {v.max():.3f} (rank {int(v.argmax())}) -> {1e3 / v.max():.0f} frames/s "
          f"for the rank set at the pace of its slowest rank")
    m = np.array([o["ms_med"] for o in rows])
    print(f"  median segments: mean {m.mean():.3f}, MAX {m.max():.3f} (rank {int(m.argmax())}) -> {1e3 / m.max():.0f} frames/s")
    if not args.ghosts and os.path.exists(ghosts):
        os.remove(ghosts)
    raise SystemExit(0)
for _try in range(8):     # (a free port can be taken between the probe and the store's listen: try another)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    try:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        break
    except Exception as e:
        if "EADDRINUSE" not in str(e) or _try == 7:
            raise
POOL = 64
PREROLL = args.preroll if args.preroll is not None else (230 if args.scene == "sweep" else 30)
if args.scene == "sweep":
    # the room sweep of sequence.py (a camera that turns and walks): frames 0 .. PREROLL + POOL of it, then the pool cycles
    from bnv_fusion_amd import sequence
    dims, voxel, scale = sequence.DIMS[args.grid]
    frames = list(sequence.sweep_frames(range(PREROLL + POOL), scale=scale, device="cuda:0"))
else:
    dims, voxel = synthetic.GRID_DIMS[args.grid]
    frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)}
              for t in range(PREROLL + POOL)]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel, tiny_cuda=args.checkpoint == "tcnn")
SHARD_KW = {k: v for k, v in (("ownership", args.ownership), ("block_log2", args.block_log2), ("axis", args.axis)) if v is not None}
lib = _lib.load()
if args.reserve:
    _lib.require_device(0)          # (the option is checked against the device's CU count)
    _lib.check(lib.bnv_set_option(b"reserve_cus", args.reserve), "reserve_cus")
if os.environ.get("BNV_HALF_TAIL") is not None:      # A/B: the table kernel's half-tile tail
    _lib.check(lib.bnv_set_option(b"half_tail", int(os.environ["BNV_HALF_TAIL"])), "half_tail")
ranks_i32 = torch.arange(W, dtype=torch.int32, device="cuda:0")


def ev(stream):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream)
    return e


def record(path):
    """All W shards in this process, lock step, real exchange (the stack of the W send blocks); keeps every frame's
    blocks and every rank's work, and runs the single volume beside them."""
    os.environ["BNV_PERSISTENT_TABLES"] = "0"     # the counts of this pass are the FULL work of a frame, like the single
    shards = [D.HipShardBackend(np.array([dims] * 3), voxel, model, r, W, capacity=1 << 21, device="cuda:0", tsdf=False,
                                n_slots=2, **SHARD_KW) for r in range(W)]     # volume's beside it: their ratio is the halo
    for b in shards:
        b.inputs_resident, b.copy_results = True, False
    model.shard = (0, 1, 3)
    single = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=1 << 21, device="cuda:0", tsdf=False)
    single.frame_pipe = False
    blocks_of, caps, work = [], [], []
    single_evals, single_vox = [], []
    t0 = time.perf_counter()
    with torch.no_grad():
        for t, fr in enumerate(frames):
            decode = t >= PREROLL
            model.shard = (0, 1, 3)
            if decode:
                c, _ = single.fuse_and_decode(fr)
                single_evals.append(int(single.volume.last_lattice_evals()[0]) if c is not None else 0)
                single_vox.append(0 if c is None else len(c))
            else:
                single.integrate(fr)
                single_evals.append(0)
                single_vox.append(0)
            frs = [b.encode(fr) for b in shards]
            bounds = [b.bound(f) for b, f in zip(shards, frs)]
            assert len(set(bounds)) == 1, bounds
            cap = shards[0].exchange_capacity(bounds[0])
            sends = [b.upsert(f, cap, decode) for b, f in zip(shards, frs)]
            blk = torch.stack(sends).clone() if cap else None
            w = []
            for b, f in zip(shards, frs):
                if cap:
                    b.install(f, blk.view(-1), cap)
                h = b.finish(f, b.decode(f) if decode else None, 0)
                c, _ = b.result(h)
                w.append((0 if c is None else len(c), b.last_owned_pairs, b._last_evals))
            blocks_of.append(None if blk is None else blk.cpu())
            caps.append(cap)
            work.append(w)
    torch.cuda.synchronize()
    work = np.array(work, dtype=np.float64)              # [frame, rank, (voxels, pairs, evaluations)]
    tail = slice(PREROLL + 8, None)                           # the frames the replay times
    ev = work[tail, :, 2]
    se = np.array(single_evals[PREROLL + 8:], dtype=np.float64)
    sent = np.array([[int(b.view(W, -1, D.REC_WORDS)[r, 0, 0]) if b is not None else 0 for r in range(W)] for b in blocks_of[PREROLL + 8:]], dtype=np.float64)
    emitted = work[tail, :, 0].sum(1)
    print(f"record pass: world {W}, {args.grid}^3, scene {args.scene}, ownership {shards[0].ownership}, blocks "
          f"{1 << shards[0].block_log2}^3" + (f", bands along axis {shards[0].axis}" if shards[0].ownership == "region" else "")
          + f": {len(frames)} frames in {time.perf_counter() - t0:.1f} s")
    print(f"  SDF-MLP evaluations per frame (pool frames): single volume {se.mean():,.0f}; per rank "
          + " ".join(f"{v:,.0f}" for v in ev.mean(0)))
    print(f"  sum over ranks / single = {ev.sum(1).mean() / se.mean():.3f}   max / mean per frame = "
          f"{(ev.max(1) / ev.mean(1)).mean():.3f}   slowest rank / (single / world) = {(ev.max(1) / (se / W)).mean():.3f}")
    print(f"  pairs max / mean = {(work[tail, :, 1].max(1) / work[tail, :, 1].mean(1)).mean():.3f}   boundary records / emitted "
          f"voxels = {(sent.sum(1) / np.maximum(emitted, 1)).mean():.3f}   records sent per rank {sent.mean():,.0f}   "
          f"all-gather {W * (np.mean(caps[PREROLL + 8:]) + 1) * 48 / 1e6:.2f} MB per rank and frame")
    torch.save({"blocks": blocks_of, "caps": caps, "work": work, "single_evals": single_evals, "world": W,
                "ownership": shards[0].ownership, "block_log2": shards[0].block_log2, "axis": shards[0].axis,
                "scene": args.scene, "grid": args.grid, "preroll": PREROLL}, path)


if args.record:
    record(args.record)
    dist.destroy_process_group()
    raise SystemExit(0)
GH = None
if args.ghosts:
    GH = torch.load(args.ghosts, weights_only=False)
    assert GH["world"] == W and GH["scene"] == args.scene and GH["grid"] == args.grid and GH["preroll"] == PREROLL
    SHARD_KW.update(ownership=GH["ownership"], block_log2=GH["block_log2"], axis=GH["axis"])
    GH["dev"] = [None if b is None else b.cuda() for b in GH["blocks"]]


def price(rank, latency):
    be = D.HipShardBackend(np.array([dims] * 3), voxel, model, rank, W, capacity=1 << 21, device="cuda:0", tsdf=True,
                           n_slots=max(4, args.in_flight + 2 + args.ahead),
                           encoder_workgroups=args.encoder_wgs, **SHARD_KW)
    be.inputs_resident = True
    be.copy_results = False
    torch.cuda.synchronize()
    stats = {"waits": 0, "recv": 0, "own": 0, "evals": 0, "pairs": 0, "enq": 0.0, "n": 0}
    HOST = {k: 0.0 for k in ("begin", "bound", "upsert", "all_gather", "simulate", "finish")}
    TRACE = []

    def exchange(f, send, cap, t):
        one = be.recv_buffer(W * send.numel())
        blocks = one.view(W, cap + 1, D.REC_WORDS)
        if GH is not None:
            # the other ranks' REAL blocks of this frame (record pass), one device copy = the payload the collective
            # would land; then the collective call itself (1-rank group) lands this rank's own block over its slot
            assert GH["caps"][t] == cap, (t, GH["caps"][t], cap)
            one.copy_(GH["dev"][t].view(-1))
            if args.exchange == "on_stream":
                blocks[rank].reshape(-1).copy_(send)
            else:
                dist.all_gather_into_tensor(blocks[rank].reshape(-1), send)
            t4 = time.perf_counter()
        else:
            dist.all_gather_into_tensor(blocks[rank].reshape(-1), send)
            t4 = time.perf_counter()
            blocks[:] = blocks[rank].clone()                      # (round 4's stand-in: copies of the own block)
            blocks[:, 0, 1] = ranks_i32
        if args.exchange_delay > 0:
            _lib.check(lib.bnv_probe_spin(1, int(args.exchange_delay * 2100), _lib.stream_ptr()), "bnv_probe_spin")
        be.install(f, one, cap)
        stats["recv"] += one.numel() * 4
        return t4

    PRE = [None]      # (frame dict, ShardFrame) whose encode was enqueued ahead
    TL_ON = [False]

    def begin(fr, nxt):
        """This frame's ShardFrame (begun now, or ahead by the previous call) and, with --ahead, the next frame's
        encode enqueued BEFORE the host waits for this frame's bound."""
        if PRE[0] is not None:
            assert PRE[0][0] is fr
            f, PRE[0] = PRE[0][1], None
        else:
            f = be.encode(fr)
        if args.ahead and nxt is not None:
            PRE[0] = (nxt, be.encode(nxt))
        return f

    def enqueue(fr, decode=True, nxt=None, t=-1):
        if args.trace and be.pipe is not None:
            E, M = be.pipe.enc, be.pipe.main
            e0 = ev(E); f = begin(fr, nxt); e1 = ev(E)
            cap = be.exchange_capacity(be.bound(f))
            m0 = ev(M); send = be.upsert(f, cap, decode); m1 = ev(M)
            if cap:
                exchange(f, send, cap, t)
            m2 = ev(M)
            h = be.finish(f, be.decode(f) if decode else None, 0)
            m3 = ev(M)
            TRACE.append((e0, e1, m0, m1, m2, m3))
            return h
        t0 = time.perf_counter()
        f = begin(fr, nxt)
        t1 = time.perf_counter()
        bound = be.bound(f); stats["waits"] += 1                     # the frame's one host wait
        t2 = time.perf_counter()
        cap = be.exchange_capacity(bound)
        send = be.upsert(f, cap, decode)
        t3 = t4 = t5 = time.perf_counter()
        if cap:
            t4 = exchange(f, send, cap, t)
            t5 = time.perf_counter()
        h = be.finish(f, be.decode(f) if decode else None, 0)
        t6 = time.perf_counter()
        for k, d in zip(HOST, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
            HOST[k] += d
        stats["enq"] += t6 - t0 - (t5 - t4)                          # (without the simulation of the other ranks)
        return h

    TL = []

    def collect(h):
        c, s = be.result(h)
        if TL_ON[0]:
            a = (C.c_float * 11)()
            _lib.check(lib.bnv_frame_timeline(be.pipe._h, h.slot, a), "bnv_frame_timeline")
            TL.append(np.array(a[:], dtype=np.float64))
        stats["own"] += 0 if c is None else len(c)
        stats["evals"] += be._last_evals
        stats["pairs"] += be.last_owned_pairs
        stats["n"] += 1

    def run(idx, in_flight, decode=True):
        pend = []
        idx = list(idx)
        for j, t in enumerate(idx):
            while len(pend) >= in_flight:
                collect(pend.pop(0))
            pend.append(enqueue(frames[t], decode, frames[idx[j + 1]] if j + 1 < len(idx) else None, t))
        while pend:
            collect(pend.pop(0))

    out = {}
    with torch.no_grad():
        run(range(PREROLL), 2, decode=False)
        run(range(PREROLL, PREROLL + 8), 2)
        idx = [PREROLL + (i % POOL) for i in range(args.frames)]
        for k in stats:
            stats[k] = 0
        for k in HOST:
            HOST[k] = 0.0
        lib.bnv_profile_enable(1)
        # timed in SEGMENTS (each ends with a device synchronisation): the figure of a run is the whole stretch, the
        # median segment says whether a hiccup of the box (another tenant, a clock dip) sits in it
        SEG = 10 if args.frames >= 1000 else 1
        seg_ms = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(SEG):
            part = idx[k * len(idx) // SEG: (k + 1) * len(idx) // SEG]
            torch.cuda.synchronize(); s0 = time.perf_counter()
            run(part, args.in_flight)
            torch.cuda.synchronize(); seg_ms.append(1e3 * (time.perf_counter() - s0) / max(len(part), 1))
        dt = time.perf_counter() - t0
        ms, cnt = (C.c_double * 4)(), (C.c_int64 * 4)()
        lib.bnv_profile_read(ms, cnt)
        lib.bnv_profile_enable(0)
        n = stats["n"]
        out = {"rank": rank, "ms": 1e3 * dt / n, "ms_med": float(np.median(seg_ms)), "ms_worst": float(np.max(seg_ms)),
               "own": stats["own"] / n, "pairs": stats["pairs"] / n,
               "evals": stats["evals"] / n, "enc_ms": ms[0] / max(cnt[0], 1), "tab_ms": ms[1] / max(cnt[1], 1),
               "host_ms": 1e3 * stats["enq"] / n}
        if GH is not None:
            # what the record pass (all W shards, real exchange) counted for this rank on the same pool frames: the
            # replay must do the same work (the live set only grows a little while the pool cycles)
            out["evals_rec"] = float(np.mean(GH["work"][PREROLL + 8:, rank, 2]))
            out["single_evals"] = float(np.mean(GH["single_evals"][PREROLL + 8:]))
        print(f"rank {rank} of a simulated world of {W}, {args.grid}^3, 640x480, {args.checkpoint} networks, {n} frames, "
              f"{args.in_flight} in flight, {args.reserve} CUs reserved, ownership {be.ownership}, "
              + (f", + {args.exchange_delay:.0f} us spin behind the stand-in all-gather" if args.exchange_delay else "") + ":")
        print(f"  pipelined wall clock  {1e3 * dt / n:.3f} ms per frame  -> {n / dt:.0f} frames/s for the rank set if every "
              f"rank keeps this pace" + (f"  ({SEG} segments: median {np.median(seg_ms):.3f}, slowest {np.max(seg_ms):.3f} ms)"
                                         if SEG > 1 else ""))
        print(f"  host enqueue time     {1e3 * stats['enq'] / n:.3f} ms per frame (includes the bound wait); host waits per "
              f"frame: {stats['waits'] / n:.2f}")
        print("  host time per frame by phase (ms): " + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in HOST.items())
              + "  ('simulate' = this tool's stand-in for the other ranks' blocks, not part of a real rank's frame)")
        print(f"  MLP kernels (HIP events, overlapping streams): point encoder {out['enc_ms']:.3f} ms, "
              f"lattice table {out['tab_ms']:.3f} ms")
        pp = be.pipe
        print("  pipeline streams verified concurrent with the main stream and with one another: " + ", ".join(
            f"{n} {getattr(st, 'bnv_concurrent', None)}" for n, st in (("encode", pp.enc), ("front", pp.front),
                                                                     ("blend", pp.blend)) if st is not None)
              + f"; encoder workgroups {pp.encoder_workgroups}")
        print(f"  voxels owned per frame {out['own']:.0f}; (point, corner) pairs encoded {out['pairs']:.0f}; SDF-MLP "
              f"evaluations {out['evals']:.0f}" + (f" (record pass, all {W} shards with the real exchange: {out['evals_rec']:.0f})"
                                                  if GH is not None else " (NO real ghost rows: --ghosts / --all-ranks)")
              + f"; bytes received per frame {stats['recv'] / n / 1e6:.2f} MB ({W} blocks)")
        if args.timeline:
            torch.cuda.synchronize()
            _lib.check(lib.bnv_frame_pipe_timeline_enable(be.pipe._h, 1), "timeline")
            run(idx[:8], args.in_flight)                      # (frames begun before the switch carry no marks)
            TL_ON[0] = True
            TL.clear()
            run(idx[8: 8 + args.timeline], args.in_flight)
            TL_ON[0] = False
            _lib.check(lib.bnv_frame_pipe_timeline_enable(be.pipe._h, 0), "timeline")
            T = np.array(TL) * 1e3                            # us
            names = ["front starts", "bound copied", "encoder starts", "encoder done", "finalize done", "upsert starts",
                     "upsert done", "finish starts", "installed", "table done", "blend done"]
            print(f"  stage timeline, GPU timestamps of {len(T)} pipelined frames (us; marker events cost a few us per frame):")
            k0 = len(T) // 2
            base = T[k0][0]
            for k in range(k0, min(k0 + 5, len(T))):
                print("    frame %d: " % (k - k0) + "  ".join(f"{n.split()[0][:5]}.{n.split()[-1][:5]} {T[k][i] - base:7.1f}" for i, n in enumerate(names)))
            d = lambda a, b: float(np.nanmean(T[4:, a] - T[4:, b]))       # noqa: E731
            x = lambda a, b: float(np.nanmean(T[5:, a] - T[4:-1, b]))     # noqa: E731  (frame t+1's point a - frame t's point b)
            print(f"    durations: front end {d(1, 0):.1f}, encoder {d(3, 2):.1f}, finalize {d(4, 3):.1f}, upsert {d(6, 5):.1f}, "
                  + f"exchange (host-enqueued) {d(7, 6):.1f}, install {d(8, 7):.1f}, marking + table {d(9, 8):.1f}, blend + read-back {d(10, 9):.1f}")
            print(f"    waits: bound copied -> encoder starts {d(2, 1):.1f}, finalize done -> upsert starts {d(5, 4):.1f}")
            print(f"    across frames: cycle (table done to table done) {x(9, 9):.1f}; table(t) done -> upsert(t+1) starts {x(5, 9):.1f}; "
                  f"encoder(t+1) starts - table(t) done {x(2, 9):.1f}; encoder(t+1) done - table(t) done {x(3, 9):.1f}; "
                  f"front(t+1) starts - table(t) done {x(0, 9):.1f}; upsert(t+1) done -> table(t+1) starts is inside 'marking + table'")
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
        if latency:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run(idx[:60], 1)
            torch.cuda.synchronize(); dt1 = time.perf_counter() - t0
            print(f"  one frame at a time   {1e3 * dt1 / 60:.3f} ms per frame (latency of a frame incl. the host round trips)")
    del be
    torch.cuda.synchronize()
    return out


gc.collect()
gc.disable()       # as bench.py: a full collection with torch imported takes ~40 ms
res = price(args.rank, latency=not args.no_latency)
if args.json:
    import json
    print(json.dumps(res))
dist.destroy_process_group()
