#!/usr/bin/env python3
"""Benchmark of the hot path: depth frames/sec fused + decoded, 640x480 @ 256^3 grid
(BASELINE.json metric; workload definition in SURVEY.md section 8d / DESIGN.md section 5).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one synthetic 640x480 frame: uint16 depth image -> points + normals (GPU front end) ->
encode_pointcloud + _integrate into the persistent volume (+ TSDF side fusion) (fused) + SDF decode of the
3x3x3 lattice of every voxel that encode returned (decoded).  Inputs are resident in HBM before the timed
region.  N > 1 is launched by torch.distributed.run, one rank per GPU over RCCL: by default frame-parallel
(ranks encode / decode different frames of a batch, replicated volume, one all-gather of encoded voxels per
batch); --parallelism spatial shards the active-voxel set by spatial hash and exchanges corner-voxel SDF tables
per frame (bnv_fusion_amd/distributed.py).  In frame-parallel mode one step is one batch
of N consecutive frames of the same stream (one per rank; weak scaling), `value` counts all of them.
The timed region runs with the cyclic garbage collector disabled and a volume that does not grow inside it
(timing hygiene); BNV_BENCH_DEBUG=1 prints per-frame enqueue times to stderr.
Rank 0 prints ONE JSON line: metric / value plus `roofline` (dominant kernel, timed alone), `kernels`,
`parity` (spot check against the oracle), `other_mlp_modes`, `cpu_baseline` (the oracle on the host cores).
"""
import argparse
import ctypes as C
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PAIR = 2 * (6 * 128 + 128 * 128 + 128 * 128 + 128 * 8)          # 69,120  point encoder
FLOP_PER_EVAL = 2 * (17 * 256 + 3 * 256 * 256 + 256)                      # 402,432 SDF MLP
FLOP_PER_PAIR_TCNN = 2 * (16 * 64 + 2 * 64 * 64 + 64 * 16)                # 20,480  tcnn encoder
FLOP_PER_EVAL_TCNN = 2 * (32 * 64 + 2 * 64 * 64 + 64 * 16)                # 22,528  tcnn decoder
PEAK_TFLOPS = {0: 157.3, 1: 2500.0, 2: 2500.0, 3: 2500.0}   # dense MFMA peaks, MI355X_MICROARCH.md: f32-in / f16-in
MODE_NAME = {0: "fp32_exact", 1: "split_f16", 2: "tcnn_f16", 3: "f16_operands"}
MFMA_PER_PRODUCT = {0: 1, 1: 3, 2: 1, 3: 1}
DTYPE = {0: "f32 (v_mfma_f32_32x32x2_f32)",
         1: "f32 operands split into f16 hi+lo, 3 products on v_mfma_f32_32x32x16_f16, f32 accumulate",
         2: "f16 weights/activations (tiny-cuda-nn FullyFusedMLP layout), f32 accumulate",
         3: "fp32 checkpoint, operands rounded to f16, 1 product on v_mfma_f32_32x32x16_f16, f32 accumulate"}


# Row capacity of the volume.  The tables grow on demand like the reference's Open3D map, but a growth step (sync +
# re-allocation + re-hash, ~40 ms) is a one-off that must not land in a 70 ms timed region: the host-side bound that
# triggers it counts the worst-case reservations of the frames in flight (up to 3 x 307,201 rows at 640x480 on top of
# ~350,000 known rows), so 2^20 rows were sometimes not enough -- and whether the step fell into the timed region
# depended on how far the host happened to run ahead.
CAPACITY = 1 << 22


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/r01_pmc_summary.csv; separate --pmc runs): 2 x FETCH_SIZE + WRITE_SIZE, in KB
    units, FETCH doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.csv")
    if not os.path.exists(path):
        return None
    fetch = write = None
    for line in open(path):
        if kernel_substr in line and "k_lattice_table_h<1>" not in line:
            parts = line.strip().rsplit(",", 3)
            if parts[1] == "FETCH_SIZE":
                fetch = float(parts[2])
            elif parts[1] == "WRITE_SIZE":
                write = float(parts[2])
    if fetch is None or write is None:
        return None
    return (2.0 * fetch + write) * 1024.0


def cpu_baseline(depth_mm, intr, T_wc, grid, n_decode_voxels=1500):
    """The oracle (PyTorch-CPU restatement of the reference) timed on this box's host cores on a
    bounded sample: one full frame depth -> points + encode + integrate, and the lattice decode of
    ``n_decode_voxels`` voxels scaled to the frame's voxel count."""
    from oracle import bnv_oracle as orc           # checker / baseline only
    from bnv_fusion_amd import synthetic
    dims, voxel = synthetic.GRID_DIMS[grid]
    sd = orc.load_weights(os.path.join(ROOT, "bnv_fusion_amd", "weights", "pointnet_fp32.npz"))
    vol = orc.OracleSparseVolume(8, voxel, np.array([dims] * 3), 8)
    # thread count: the fastest of a few candidates on a 40k-point encode (all cores is NOT the
    # fastest on a many-core host), so the baseline is not handicapped
    t0 = time.perf_counter()
    pts_np = orc.depth_to_input_pts(depth_mm.astype(np.float64) / 1000.0, intr, T_wc)
    t_front = time.perf_counter() - t0
    frame0 = pts_np.astype(np.float32)[None]
    probe = torch.from_numpy(frame0)[:, :40000]
    best = None
    for th in sorted({min(c, os.cpu_count() or 1) for c in (8, 16, 32, 64, 128, 256)}):
        torch.set_num_threads(th)
        with torch.no_grad():
            orc.encode_pointcloud(sd, probe, vol.n_xyz, vol.min_coords, vol.max_coords, voxel)
            t0 = time.perf_counter()
            orc.encode_pointcloud(sd, probe, vol.n_xyz, vol.min_coords, vol.max_coords, voxel)
            dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, th)
    threads = best[1]
    torch.set_num_threads(threads)
    pts = torch.from_numpy(frame0)
    with torch.no_grad():
        t0 = time.perf_counter()
        f, c, ids, g, n = orc.encode_pointcloud(sd, pts, vol.n_xyz, vol.min_coords, vol.max_coords, voxel)
        t_enc = time.perf_counter() - t0
        t0 = time.perf_counter()
        orc.integrate(vol, g, f, c)
        t_int = time.perf_counter() - t0
        sel = g[:: max(1, len(g) // n_decode_voxels)][:n_decode_voxels]
        t0 = time.perf_counter()
        vol.decode_pts(orc.lattice_coords(sel.numpy()), sd, None, is_coords=True, query_tensor=False)
        t_dec = (time.perf_counter() - t0) * len(g) / len(sel)
    total = t_front + t_enc + t_int + t_dec
    return {"value": 1.0 / total, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": (f"oracle (PyTorch-CPU fp32 restatement, {threads} threads; numpy float64 front end): 1 full "
                       f"640x480 frame depth->points {t_front:.2f}s + encode {t_enc:.2f}s + integrate {t_int:.2f}s + "
                       f"lattice decode of {len(sel)} of {len(g)} voxels scaled to the frame = {t_dec:.2f}s"),
            "front_end_s": t_front, "encode_s": t_enc, "integrate_s": t_int, "decode_s_scaled": t_dec}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=256, choices=[128, 256, 512])
    ap.add_argument("--preroll", type=int, default=30,
                    help="frames fused (untimed setup) before warm-up so that voxel weights reach "
                         "min_pts_in_grid and the decode mask is live (SURVEY.md section 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mlp-mode", type=int, default=1, choices=[0, 1, 3],
                    help="1 (default): split-f16 operands on the f16 MFMA; 0: exact fp32 MFMA")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the short run in the other MLP mode")
    ap.add_argument("--checkpoint", default="fp32", choices=["fp32", "tcnn"],
                    help="fp32: pointnet.ckpt networks (oracle-pinned; the headline); tcnn: the reference's default "
                         "tiny-cuda-nn fp16 networks (pointnet_tcnn.ckpt)")
    ap.add_argument("--input", default="depth", choices=["depth", "points"],
                    help="'depth' (default): a step starts from the uint16 depth image resident in HBM and runs the "
                         "GPU front end (unprojection + normals); 'points': from precomputed input_pts")
    ap.add_argument("--sync-frames", action="store_true",
                    help="1 GPU: use the synchronous per-frame API (one host sync mid-frame) instead of the "
                         "pipelined fuse_and_decode_async")
    ap.add_argument("--no-stream-overlap", action="store_true",
                    help="1 GPU: enqueue a frame's encode on the main stream instead of a second HIP stream")
    ap.add_argument("--parallelism", default="frame", choices=["frame", "spatial"],
                    help="N > 1: 'frame' = ranks encode/decode different frames of a batch, replicated volume, one "
                         "all-gather per batch (throughput scaling); 'spatial' = voxels sharded by spatial hash, "
                         "table all-gather per frame")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    # one rank per GPU; BNV_DIST_BACKEND=gloo lets several ranks share a GPU for functional testing
    backend = os.environ.get("BNV_DIST_BACKEND", "nccl")
    local_dev = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_dev)
    dev = f"cuda:{local_dev}"
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            # RCCL's stream at high priority: the exchange kernels are short and on the path of the next batch;
            # they should take a CU as soon as one of the (CU-filling) compute kernels releases it
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", device_id=torch.device(dev), pg_options=opts)
        else:
            dist.init_process_group(backend)

    import bnv_fusion_amd as bnv
    from bnv_fusion_amd import synthetic, _lib

    dims, voxel = synthetic.GRID_DIMS[args.grid]
    tcnn = args.checkpoint == "tcnn"
    model = bnv.load_pretrained(device=dev, voxel_size=voxel, tiny_cuda=tcnn)
    if tcnn:
        args.no_alt_mode = True
        args.mlp_mode = 2
    frame_parallel = world > 1 and args.parallelism == "frame"
    if frame_parallel:
        from bnv_fusion_amd.distributed import FrameParallelNeuralMap
        nm = FrameParallelNeuralMap(np.array([dims] * 3), voxel, model, device=dev, tsdf=(args.input == "depth"),
                                    capacity=CAPACITY)
    elif world > 1:
        from bnv_fusion_amd.distributed import ShardedNeuralMap
        nm = ShardedNeuralMap(np.array([dims] * 3), voxel, model, device=dev)
    else:
        nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=CAPACITY, device=dev,
                           tsdf=(args.input == "depth"))
        nm.overlap_encode = not args.no_stream_overlap

    # ---- synthetic inputs, resident in HBM before anything is timed -----------------------------
    # frames per step: one frame on one GPU; in frame-parallel mode a step is one batch = one frame PER RANK
    # (weak scaling: per-GPU work per step is fixed, `value` counts the frames of all ranks)
    fpu = world if frame_parallel else 1
    n_frames = args.preroll + (args.warmup + args.steps) * fpu
    depth_host = [synthetic.depth_u16(t) for t in range(n_frames)]
    intr = synthetic.intrinsics()
    if args.input == "depth":
        frames = [{"depth": torch.from_numpy(d).to(dev), "intr_mat": intr, "T_wc": synthetic.pose(t)}
                  for t, d in enumerate(depth_host)]
    else:
        frames = [{"input_pts": torch.from_numpy(synthetic.depth_to_input_pts(
            d.astype(np.float64) / 1000.0, intr, synthetic.pose(t)).astype(np.float32)[None]).to(dev)}
            for t, d in enumerate(depth_host)]
    n_points = int((depth_host[0] > 0).sum())

    def run_frames(first, count, decode=True, collect=None):
        """Processes frames [first, first+count) in order; returns this rank's last (coords, sdf)."""
        last = (None, None)
        if frame_parallel:
            # batches of `world` consecutive frames, software-pipelined (batch k+1's encode + all-gather are
            # enqueued before batch k's integrate + decode); only the last handle is read back on the host
            batches = [frames[t0: min(t0 + world, first + count)] for t0 in range(first, first + count, world)]
            handle = None
            for handle in nm.process_stream(batches, decode=decode):
                pass
            nm.flush()
            if handle is not None:
                out = handle.result()
                if out[0] is not None:
                    last = out
        elif world == 1 and not args.sync_frames:
            # software pipeline: frame t is enqueued before frame t-1's result is collected, so the GPU
            # never waits for the host (each result() waits on that frame's own event only)
            pending = None
            _dbg = [] if os.environ.get("BNV_BENCH_DEBUG") else None
            for t in range(first, first + count):
                _a = time.perf_counter()
                h = nm.fuse_and_decode_async(frames[t], decode=decode)
                if _dbg is not None:
                    _dbg.append(time.perf_counter() - _a)
                if pending is not None:
                    r = pending.result()
                    if collect is not None:
                        collect(r)
                pending = h
            last = pending.result()
            if _dbg:
                print("enqueue ms:", " ".join(f"{1e3*x:.2f}" for x in _dbg), file=sys.stderr)
            if collect is not None:
                collect(last)
        else:
            for t in range(first, first + count):
                last = nm.fuse_and_decode(frames[t]) if decode else (nm.integrate(frames[t]), None)
                if collect is not None:
                    collect(last)
        return last

    run_frames(0, args.preroll, decode=False)           # setup: make the decode mask live

    lib = _lib.load()

    def timed(mode, first, steps, warm):
        """`warm` untimed steps, then times exactly `steps` steps (`fpu` frames each) from frame index `first`, in
        MLP mode `mode`."""
        if mode != 2:
            bnv.set_mlp_mode(mode)
        run_frames(first - warm * fpu, warm * fpu)
        lib.bnv_profile_enable(1)
        table_rows, n_vox = [], []
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # the cyclic garbage collector stays out of the timed region (as timeit does): with torch imported a full
        # collection takes ~40 ms, and one landed on the third timed frame of every process but the first on a box
        # (350 instead of 540 frames/s over 40 frames; found with per-frame enqueue times, BNV_BENCH_DEBUG=1)
        gc.collect()
        gc.disable()
        _ms0 = torch.cuda.memory_stats() if os.environ.get("BNV_BENCH_DEBUG") else None
        t0 = time.perf_counter()
        def collect(res):
            c, _ = res
            n_vox.append(0 if c is None else int(c.shape[0]))

        if frame_parallel:
            coords, sdf = run_frames(first, steps * fpu)
            table_rows.append(nm.volume.last_lattice_evals().clone() if coords is not None
                              else torch.zeros(1, dtype=torch.int32, device=dev))
            n_vox.append(0 if coords is None else int(coords.shape[0]))
        elif world > 1:
            for t in range(first, first + steps):
                coords, sdf = nm.fuse_and_decode(frames[t])
                table_rows.append((nm.volume.last_lattice_table_rows() * 27).clone())
                n_vox.append(0 if coords is None else int(coords.shape[0]))
        else:
            coords, sdf = run_frames(first, steps, collect=collect)
            table_rows.append(nm.volume.last_lattice_evals().clone())
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        gc.enable()
        if _ms0 is not None:
            _ms1 = torch.cuda.memory_stats()
            print("timed region: device allocs", _ms1["num_device_alloc"] - _ms0["num_device_alloc"], "device frees",
                  _ms1["num_device_free"] - _ms0["num_device_free"], "reserved MB",
                  _ms1["reserved_bytes.all.current"] >> 20, "retries", _ms1["num_alloc_retries"], file=sys.stderr)
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        prof_ms = (C.c_double * 4)()
        prof_n = (C.c_int64 * 4)()
        lib.bnv_profile_read(prof_ms, prof_n)
        lib.bnv_profile_enable(0)
        rows = torch.stack(table_rows).cpu().numpy().reshape(-1)
        live = float((sdf != voxel).float().mean()) if sdf is not None and sdf.numel() else 0.0
        dec_ms = prof_ms[1] / max(prof_n[1], 1)
        enc_ms = prof_ms[0] / max(prof_n[0], 1)
        dec_flop = float(rows.mean()) * (FLOP_PER_EVAL_TCNN if mode == 2 else FLOP_PER_EVAL)
        enc_flop = 8.0 * n_points * (FLOP_PER_PAIR_TCNN if mode == 2 else FLOP_PER_PAIR)
        return {"elapsed": elapsed, "steps": steps, "fps": steps * fpu / elapsed, "rows": float(rows.mean()),
                "n_vox": float(np.mean(n_vox)), "live": live, "dec_ms": dec_ms, "enc_ms": enc_ms,
                "dec_tflops": dec_flop / (dec_ms * 1e-3) / 1e12 if dec_ms else 0.0,
                "enc_tflops": enc_flop / (enc_ms * 1e-3) / 1e12 if enc_ms else 0.0, "dec_flop": dec_flop,
                "coords": coords, "sdf": sdf}

    first = args.preroll + args.warmup * fpu
    main_run = timed(args.mlp_mode, first, args.steps, args.warmup)
    elapsed = main_run["elapsed"]
    # parity spot check of a configuration against the oracle (40 voxels of its last frame): the SDF decoded by
    # the GPU from the GPU's own volume vs the oracle's decode of the same volume values
    def parity_check(run):
        if not ((world == 1 or frame_parallel) and rank == 0 and run["coords"] is not None):
            return None
        from oracle import bnv_oracle as orc           # checker only
        sd = orc.load_weights(os.path.join(ROOT, "bnv_fusion_amd", "weights", "pointnet_fp32.npz"))
        geo = None
        if tcnn:
            geo = orc.tcnn_geo_forward(orc.load_weights(os.path.join(
                ROOT, "bnv_fusion_amd", "weights", "pointnet_tcnn.npz"))["nerf.model.params"])
        g = run["coords"]
        sel = torch.randperm(len(g), generator=torch.Generator().manual_seed(0))[:40].to(g.device)
        pick = g[sel].cpu()
        off = torch.tensor([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)])
        nbr = torch.unique((pick[:, None, :] + off[None]).reshape(-1, 3), dim=0)
        fo, wo, _ = nm.volume.query(nbr.to(dev))
        ovol = orc.OracleSparseVolume(8, voxel, np.array([dims] * 3), 8)
        present = wo[:, 0].cpu() > 0
        ovol.insert(nbr[present], fo.cpu()[present], wo.cpu()[present], torch.zeros(int(present.sum()), 1))
        ref = ovol.decode_pts(orc.lattice_coords(pick.numpy()), sd, None, is_coords=True, query_tensor=False,
                              geo=geo)[0, :, :, 0]
        if world == 1:
            got = run["sdf"][sel].cpu()     # the very output of the last timed frame (no extra launch)
        else:                               # replicated volume has moved on: decode again from the current state
            got = nm.volume.decode_lattice(pick.to(dev), model.nerf, query_tensor=False).cpu()
        return {"sdf_max_abs_err_vs_oracle": float((got - ref).abs().max()), "tolerance": 1e-4,
                "oracle": "fp16 restatement of the tcnn layout (parity unpinned)" if tcnn else "pinned fp32 oracle",
                "mask_decisions_equal": bool(torch.equal(got == voxel, ref == voxel)), "voxels_checked": 40}

    parity = parity_check(main_run)          # right after the timed region: the volume is in that run's final state
    # Kernel-alone pass for the roofline: with the encode on a second stream the two MLP kernels of consecutive
    # frames share the GPU, so their event-to-event durations overlap.  A roofline needs the kernel's own
    # duration: a few of the same frames are run again with everything on one stream (and that is how the
    # committed rocprofv3 summaries are taken: --no-stream-overlap).
    kern_run, kern_note = main_run, "timed region"
    if world == 1 and getattr(nm, "overlap_encode", False):
        nm.overlap_encode = False
        # (the CPU-side parity check above left the GPU idle for seconds: same cold start + warm-up as the timed region)
        kern_run = timed(args.mlp_mode, first, args.steps, args.warmup)
        nm.overlap_encode = True
        kern_note = (f"the same {kern_run['steps']} frames (+{args.warmup} warm-up) re-run from an idle GPU with the "
                     "encode on the main stream, so that each kernel has the GPU to itself (in the timed region the "
                     "two MLP kernels of consecutive frames overlap); that pass ran at "
                     f"{kern_run['fps']:.1f} frames/s")

    # the other arithmetic modes of the fp32 checkpoint on a few of the same frames (the volume state differs
    # only by those fusions), each with its own parity spot check
    alts = []
    if not args.no_alt_mode and world == 1 and not tcnn:
        for am in (0, 1, 3):
            if am == args.mlp_mode:
                continue
            r = timed(am, first, min(args.steps, 8), max(args.warmup, 3))
            r["mode"], r["parity"] = am, parity_check(r)
            alts.append(r)
        bnv.set_mlp_mode(args.mlp_mode)

    if rank == 0:
        fps = args.steps * fpu / elapsed
        m = args.mlp_mode
        peak = PEAK_TFLOPS[m]
        out = {
            "metric": "depth frames/sec fused+decoded, 640x480 @ 256^3 grid",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak" if (frame_parallel or world == 1) else "strong",
            "vs_baseline": None, "dtype": DTYPE[m], "data": "synthetic",
            "config": {"workload": f"synthetic 640x480 depth ({n_points} valid points/frame), {args.grid}^3 grid, "
                                   f"voxel {voxel}, {'pointnet_tcnn.ckpt (fp16 tcnn)' if tcnn else 'fp32 pointnet.ckpt'} weights; step = "
                                   + ("uint16 depth image -> points + normals (GPU front end) + "
                                      if args.input == "depth" else "")
                                   + "encode_pointcloud + _integrate + "
                                   + ("TSDF side fusion at 0.025 m + " if (args.input == "depth" and
                                                                          (world == 1 or frame_parallel)) else "")
                                   + "decode of the 3x3x3 lattice of every "
                                     "touched voxel",
                       "grid": args.grid, "voxel_size": voxel, "preroll_frames": args.preroll,
                       "frames_per_step": fpu,
                       "mlp_mode": MODE_NAME[m],
                       "voxels_per_frame": main_run["n_vox"], "sdf_values_per_frame": 27.0 * main_run["n_vox"],
                       "decode_live_fraction": main_run["live"],
                       "parallelism": ("1 GPU" if world == 1 else
                                       f"frame-parallel x{world}: one step = one batch of {world} consecutive frames, ranks encode/decode "
                                       "different frames of the batch, "
                                       "replicated volume, one RCCL all-gather of encoded voxels per batch"
                                       if frame_parallel else
                                       f"spatial-hash voxel sharding x{world} + RCCL all-gather of SDF tables per frame")},
            # dominant kernel: the lattice-table SDF MLP.  achieved = algorithmic FLOPs (402,432 per MLP
            # evaluation x evaluations per launch; the split mode issues 3 MFMA products per algorithmic
            # product, which are NOT counted) / mean kernel time from HIP events on the launch stream
            "roofline": {"bound": "mfma", "kernel": ("k_lattice_table_h" if m in (1, 3) else f"k_decode<LATTICE,{MODE_NAME[m]}>") + " (SDF MLP 17-256x4-1)",
                         "achieved": kern_run["dec_tflops"], "peak": peak, "unit": "TFLOP/s",
                         "frac": kern_run["dec_tflops"] / peak,
                         "traffic": pmc_traffic_bytes("k_lattice_table_h") if (m == 1 and world == 1 and not tcnn) else None,
                         "traffic_note": "HBM bytes/launch, rocprofv3 PMC (profiles/r01_pmc_summary.csv); "
                                         "algorithmic bytes = 40 B x evaluations",
                         "avg_kernel_ms": kern_run["dec_ms"], "flop_per_launch": kern_run["dec_flop"],
                         "mlp_evals_per_launch": kern_run["rows"],
                         "mfma_issue_frac": kern_run["dec_tflops"] * MFMA_PER_PRODUCT[m] / peak,
                         "timing": kern_note},
            "kernels": {"pointnet_scatter": {"avg_ms": kern_run["enc_ms"], "tflops": kern_run["enc_tflops"],
                                             "frac_of_peak": kern_run["enc_tflops"] / peak}},
            "parity": parity,
        }
        out["other_mlp_modes"] = [
            {"mlp_mode": MODE_NAME[a["mode"]], "dtype": DTYPE[a["mode"]], "value": a["fps"], "unit": "frames/s",
             "steps": a["steps"], "ms_per_step": 1e3 * a["elapsed"] / a["steps"], "decode_kernel_ms": a["dec_ms"],
             "decode_tflops": a["dec_tflops"], "decode_frac_of_peak": a["dec_tflops"] / PEAK_TFLOPS[a["mode"]],
             "pointnet_kernel_ms": a["enc_ms"], "pointnet_tflops": a["enc_tflops"], "parity": a["parity"]}
            for a in alts]
        if not args.no_cpu_baseline and world == 1 and not tcnn:
            out["cpu_baseline"] = cpu_baseline(depth_host[0], intr, synthetic.pose(0), args.grid)
            out["speedup_vs_cpu_baseline"] = fps / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
