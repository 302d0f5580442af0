#!/usr/bin/env python3
"""Benchmark of the hot path: depth frames/sec fused + decoded, 640x480 @ 256^3 grid
(BASELINE.json metric; workload definition in SURVEY.md section 8d / DESIGN.md section 5).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one synthetic 640x480 frame: uint16 depth image -> points + normals (GPU front end) ->
encode_pointcloud + _integrate into the persistent volume (+ TSDF side fusion) (fused) + SDF decode of the
3x3x3 lattice of every voxel that encode returned (decoded).  Inputs are resident in HBM before the timed
region.

N > 1: one rank per GPU over RCCL.  Started either by `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or as plain `python bench.py --gpus N`, which
starts the N ranks itself as a child `torch.distributed.run` BEFORE this process makes any GPU call and exits with
the child's code.  Both decompositions are measured in the same run; `value` is the one --parallelism names.
Default: 'spatial', the decomposition BASELINE.json names -- the active-voxel set sharded by spatial hash, every rank
works on every frame, one all-gather of boundary-voxel records per frame (strong scaling of one stream; a step = one
frame).  'frame': ranks encode / decode different frames of a batch, replicated volume, one all-gather of encoded
voxels per batch (weak scaling; a step = one batch of N consecutive frames, `value` counts all of them)
(bnv_fusion_amd/distributed.py).  Collectives time out (--dist-timeout) and any rank's failure
ends the job with a non-zero exit code.

The timed region runs with the cyclic garbage collector disabled and a volume that does not grow inside it
(timing hygiene); BNV_BENCH_DEBUG=1 prints per-frame enqueue times to stderr.
Rank 0 prints ONE JSON line: metric / value -- a SUSTAINED rate: the timed steps run right behind --preheat untimed
frames, so the clock has settled under the package power limit -- plus
  burst            the same steps from an idle GPU (what earlier rounds reported as `value`);
  roofline         dominant kernel of the headline arithmetic mode, timed alone (HIP events on its stream);
  fp32_exact       the same --steps in IEEE-fp32 MFMA arithmetic (the reference's precision to the letter), with its
                   own kernel-alone roofline and parity check;
  sustained        >= 1,000 frames back to back (frames/s, shader clock and package power while it ran);
  growth           frames/s from an empty volume of the reference's initial capacity (100,000 rows), growing on demand;
  sequence         a long moving-camera sweep of a room-sized 512^3 volume (bnv_fusion_amd/sequence.py): frames/s over
                   the whole sequence incl. table growth, rows reached, periodic oracle checks;
  optimize         the global optimiser at the reference's 5,000-ray configuration (run_e2e.py:111-162): steps/s, the two
                   decode_pts kernels of a ray split with their roofline fractions, forward + gradient parity against the
                   oracle's autograd, the oracle's own time;
  extract_mesh[_sweep]  NeuralMap.extract_mesh over the whole bench volume / the sequence's volume (run_e2e.py:164-167):
                   ms, table-kernel and marching-cubes figures, marching cubes against the oracle's (section 8 f-4);
  parity           GPU outputs of the last timed frame against the oracle (>= 2,000 voxels);
  other_mlp_modes  the f16-operand mode (lower precision, never the headline);
  cpu_baseline     the oracle on the host cores.
"""
import argparse
import ctypes as C
import gc
import json
import os
import socket
import subprocess
import sys
import threading
import time

# streams that share a hardware queue run in submission order (bnv_fusion_amd/streams.py): more queues, set before the
# HIP runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from bnv_fusion_amd import configure_runtime  # noqa: E402

configure_runtime()      # 8 hardware queues for the frame pipelines' streams; before the first HIP call

FLOP_PER_PAIR = 2 * (6 * 128 + 128 * 128 + 128 * 128 + 128 * 8)          # 69,120  point encoder
FLOP_PER_EVAL = 2 * (17 * 256 + 3 * 256 * 256 + 256)                      # 402,432 SDF MLP
FLOP_PER_PAIR_TCNN = 2 * (16 * 64 + 2 * 64 * 64 + 64 * 16)                # 20,480  tcnn encoder
FLOP_PER_EVAL_TCNN = 2 * (32 * 64 + 2 * 64 * 64 + 64 * 16)                # 22,528  tcnn decoder
PEAK_TFLOPS = {0: 157.3, 1: 2500.0, 2: 2500.0, 3: 2500.0}   # dense MFMA peaks, MI355X_MICROARCH.md: f32-in / f16-in
MODE_NAME = {0: "fp32_exact", 1: "split_f16", 2: "tcnn_f16", 3: "f16_operands"}
MFMA_PER_PRODUCT = {0: 1, 1: 3, 2: 1, 3: 1}
DTYPE = {0: "f32 (v_mfma_f32_32x32x2_f32)",
         1: "f32 operands split into f16 hi+lo, 3 products on v_mfma_f32_16x16x32_f16, f32 accumulate",
         2: "f16 weights/activations (tiny-cuda-nn FullyFusedMLP layout), f32 accumulate",
         3: "fp32 checkpoint, operands rounded to f16, 1 product on v_mfma_f32_16x16x32_f16, f32 accumulate"}
DECODE_KERNEL = {0: "k_decode<LATTICE, fp32_exact>", 1: "k_lattice_table_x<3>", 2: "k_lattice_table_t",
                 3: "k_lattice_table_x<1>"}
PARITY_VOXELS = 2048

# Row capacity of the volume in the timed runs.  The tables grow on demand like the reference's Open3D map, but a
# growth step (sync + re-allocation + re-hash, ~40 ms) is a one-off that must not land in a 100 ms timed region: the
# host-side bound that triggers it counts the worst-case reservations of the frames in flight (up to 3 x 307,201 rows
# at 640x480 on top of ~350,000 known rows).  What growing from the reference's 100,000 rows costs is reported
# separately (`growth`).
CAPACITY = 1 << 22
PROFILE_TAG = "r06"


def pmc_traffic(kernel_substr, evals_now):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/<tag>_pmc_summary.csv; separate --pmc runs): 2 x FETCH_SIZE + WRITE_SIZE in KB units, FETCH doubled
    as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950.  The counters cannot be collected from
    inside this process, so the figure is the profile's, with its source named next to it.  The kernel's HBM traffic
    is proportional to its evaluations per launch (a 32-B feature row and a 4-B entry read, a 4-B table entry written
    per evaluation; the weights stream from L2), so when this run's launches are up to 25 % larger or smaller than the
    profiled ones the figure is scaled by that ratio -- and says so; beyond that it is dropped."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_summary.csv")
    meta_path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_meta.json")
    if not (os.path.exists(path) and os.path.exists(meta_path)):
        return None, None
    meta = json.load(open(meta_path))
    # the profile is only this kernel's while the kernel's source is the one that was profiled (the GPU box has no
    # git history: the profile's meta file carries the SHA-256 of csrc/decode.hip as it was then)
    import hashlib
    with open(os.path.join(ROOT, "bnv_fusion_amd", "csrc", "decode.hip"), "rb") as fh:
        sha_now = hashlib.sha256(fh.read()).hexdigest()
    if meta.get("decode_hip_sha256") != sha_now:
        return None, {"file": f"profiles/{PROFILE_TAG}_pmc_summary.csv", "source_commit": meta.get("commit"),
                      "dropped": "csrc/decode.hip has changed since the PMC passes were taken (or the profile does "
                                 "not say which source it saw): re-run tools/run_profiles.sh"}
    fetch = write = None
    for line in open(path):
        if kernel_substr in line:
            parts = line.strip().rsplit(",", 3)
            if parts[1] == "FETCH_SIZE":
                fetch = float(parts[2])
            elif parts[1] == "WRITE_SIZE":
                write = float(parts[2])
    if fetch is None or write is None:
        return None, None
    prof_evals = float(meta.get("mlp_evals_per_launch", 0.0))
    src = {"file": f"profiles/{PROFILE_TAG}_pmc_summary.csv", "source_commit": meta.get("commit"),
           "mlp_evals_per_launch_in_profile": prof_evals,
           "algorithmic_bytes": "40 B x evaluations (32 B feature row + 4 B entry id read, 4 B table entry written)"}
    if not prof_evals or abs(evals_now - prof_evals) > 0.25 * prof_evals:
        src["dropped"] = "evaluations per launch of this run differ from the profiled run's by more than 25 %"
        return None, src
    ratio = evals_now / prof_evals
    if abs(ratio - 1.0) > 0.02:
        src["scaled_by_evaluations_per_launch"] = ratio
    return (2.0 * fetch + write) * 1024.0 * ratio, src


class SmiSampler:
    """Samples shader clock and package power with rocm-smi from a host thread while a pass runs."""

    def __init__(self, period=0.25):
        self.period, self.samples, self._stop, self._th = period, [], False, None
        # Under a profiler (rocprofv3 preloads a tool library that initialises the GPU before main) a child that
        # re-execs -- rocm-smi is a `#!/usr/bin/env python3` script -- is the exec-after-GPU-init this pool forbids:
        # no sampling then; otherwise the child gets an environment without any preload / profiler variables.
        env = os.environ
        self.disabled = ("rocprof" in env.get("LD_PRELOAD", "").lower()
                         or any(k.startswith(("ROCPROFILER_", "ROCP_TOOL", "ROCPROF_")) for k in env))
        self._env = {k: v for k, v in env.items()
                     if k != "LD_PRELOAD" and not k.startswith(("ROCPROFILER_", "ROCP_", "ROCPROF_", "HSA_TOOLS_"))}

    def _loop(self):
        while not self._stop and not self.disabled:
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True,
                                   text=True, timeout=5, env=self._env).stdout
                card = next(iter(json.loads(o).values()))
                sclk = pwr = None
                for k, v in card.items():
                    kl = k.lower()
                    if "sclk" in kl and "mhz" in str(v).lower():
                        sclk = float(str(v).lower().replace("(", " ").replace("mhz", " ").split()[0])
                    elif "power" in kl and "(w)" in kl:
                        try:
                            pwr = float(v)
                        except ValueError:
                            pass
                self.samples.append((time.perf_counter(), sclk, pwr))
            except Exception:
                return
            time.sleep(self.period)

    def __enter__(self):
        self._th = threading.Thread(target=self._loop, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        self._th.join(timeout=10)

    def summary(self, t0, t1):
        inside = [(c, p) for (t, c, p) in self.samples if t0 <= t <= t1]
        clk = [c for c, _ in inside if c]
        pw = [p for _, p in inside if p]
        return {"smi_samples": len(inside), "mean_sclk_mhz": float(np.mean(clk)) if clk else None,
                "mean_package_power_w": float(np.mean(pw)) if pw else None}


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
    return {"value": 1.0 / total, "unit": "frames/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
            "sample": (f"oracle (PyTorch-CPU fp32 restatement, {threads} threads -- the fastest of a probe -- on a host "
                       f"with {os.cpu_count()} cores; numpy float64 front end): 1 full "
                       f"640x480 frame depth->points {t_front:.2f}s + encode {t_enc:.2f}s + integrate {t_int:.2f}s + "
                       f"lattice decode of {len(sel)} of {len(g)} voxels scaled to the frame = {t_dec:.2f}s"),
            "front_end_s": t_front, "encode_s": t_enc, "integrate_s": t_int, "decode_s_scaled": t_dec}


def oracle_lattice_check(volume, coords, sdf, tcnn=False, n_voxels=PARITY_VOXELS, seed=0, decode_again=None):
    """SDF lattices the GPU decoded for ``n_voxels`` of ``coords`` against the CPU oracle's decode of the SAME volume
    values (the voxels' 3x3x3 neighbourhoods are read back from the GPU volume -- for a shard: own + ghost rows).
    Checker only.  -> dict like the line's `parity`."""
    from oracle import bnv_oracle as orc           # checker only
    sd = orc.load_weights(os.path.join(ROOT, "bnv_fusion_amd", "weights", "pointnet_fp32.npz"))
    geo = None
    if tcnn:
        geo = orc.tcnn_geo_forward(orc.load_weights(os.path.join(
            ROOT, "bnv_fusion_amd", "weights", "pointnet_tcnn.npz"))["nerf.model.params"])
    dev = coords.device
    voxel = volume.voxel_size
    sel = torch.randperm(len(coords), generator=torch.Generator().manual_seed(seed))[:n_voxels].to(dev)
    pick = coords[sel].cpu()
    off = torch.tensor([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)])
    nbr = torch.unique((pick[:, None, :] + off[None]).reshape(-1, 3), dim=0)
    fo, wo, _ = volume.query(nbr.to(dev))
    ovol = orc.OracleSparseVolume(8, voxel, np.asarray(volume.dimensions), 8)
    present = wo[:, 0].cpu() > 0
    ovol.insert(nbr[present], fo.cpu()[present], wo.cpu()[present], torch.zeros(int(present.sum()), 1))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        ref = ovol.decode_pts(orc.lattice_coords(pick.numpy()), sd, None, is_coords=True, query_tensor=False,
                              geo=geo)[0, :, :, 0]
    got = sdf[sel].cpu() if decode_again is None else decode_again(pick.to(dev)).cpu()
    return {"sdf_max_abs_err_vs_oracle": float((got - ref).abs().max()), "tolerance": 1e-4,
            "oracle": "fp16 restatement of the tcnn layout (parity unpinned)" if tcnn else "pinned fp32 oracle",
            "mask_decisions_equal": bool(torch.equal(got == voxel, ref == voxel)),
            "live_fraction_checked": float((ref != voxel).float().mean()),
            "voxels_checked": int(len(pick)), "sdf_values_checked": int(ref.numel())}


def _event_ms(fn, reps, stream=None):
    """Mean duration (ms) of ``fn()`` over ``reps`` calls between two HIP events on the current stream."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def optimize_entry(nm, model, frames, mlp_mode, n_iters=200):
    """SURVEY.md section 8 f-3, the global optimiser at the reference's configuration (run_e2e.py:111-162,
    fusion_pointnet_model.yaml): Adam steps on the volume features, 5,000 rays of a random key frame per step in splits
    of 1,000 rays, 20 fine + 15 coarse samples per ray.  -> steps/s ("speed on global fusion", run_e2e.py:289), the two
    kernels of a split timed alone with their roofline fractions, parity of one split against the oracle's autograd
    and the oracle's own time for that split on the host cores."""
    from bnv_fusion_amd import optimize
    from oracle import bnv_oracle as orc           # checker / baseline only
    dev = nm.volume._dev
    voxel = nm.volume.voxel_size
    nm.frames = list(frames)
    gen = torch.Generator(device=dev).manual_seed(0)
    nm.optimize(n_iters=40, last_frame=-1, generator=gen)     # warm-up (allocations, Adam state, the key frames' points)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = nm.optimize(n_iters=n_iters, last_frame=-1, generator=gen)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    vol = nm.volume
    vol.to_tensor()
    vol.features = torch.nn.Parameter(vol.features)
    f = frames[3]
    d = f["depth"]
    d = d.to(torch.float32) / 1000.0 if d.dtype in (torch.uint16, torch.int16) else d
    # ---- the step's fused launch set (round 6): all 5 splits sampled, counted, decoded forward + loss + backward in ONE
    # kernel (k_optim_step), the counts applied -- timed with HIP events on a fixed ray batch
    rays5 = optimize.sample_key_frame(d, f["intr_mat"], f["T_wc"], 5000, 3, generator=gen)
    grad5 = torch.zeros_like(vol.features)
    w_keep = vol.weights.clone()
    _, _, pred5 = optimize.ray_batch_step(vol, rays5, model.nerf, nm.truncated_units, nm.truncated_dist, 3,
                                          generator=gen, grad=grad5, return_pred=True)
    live5 = int((pred5 != voxel).sum())
    step_ms = _event_ms(lambda: optimize.ray_batch_step(vol, rays5, model.nerf, nm.truncated_units, nm.truncated_dist, 3,
                                                        generator=gen, grad=grad5), 20)
    vol.weights.copy_(w_keep)            # (the timing loop's count_optim calls are not part of the optimisation)
    # ---- one split (1,000 rays x 35 samples) through the separate forward / backward kernels (decode_pts + autograd)
    rays = optimize.sample_key_frame(d, f["intr_mat"], f["T_wc"], 1000, 3, generator=gen)
    with torch.no_grad():
        out = optimize.render_with_rays(vol, rays, model.nerf, None, nm.truncated_units, nm.truncated_dist, 3, generator=gen)
    pts = out["pts_on_rays"].detach()
    n_q = int(pts.numel() // 3)
    with torch.no_grad():
        sdf0 = vol.decode_pts(pts, model.nerf, None)
    live = int((sdf0 != voxel).sum())
    fwd_ms = _event_ms(lambda: vol.decode_pts(pts, model.nerf, None).detach(), 20)
    go = torch.ones_like(sdf0)

    def fwd_bwd():
        vol.features.grad = None
        vol.decode_pts(pts, model.nerf, None).backward(go)

    both_ms = _event_ms(fwd_bwd, 20)
    bwd_ms = max(both_ms - fwd_ms, 1e-6)
    flop_fwd = live * 8.0 * FLOP_PER_EVAL
    peak = PEAK_TFLOPS[mlp_mode]
    # ---- parity of a 6,000-query sample against the oracle's forward + autograd (CPU), and the oracle's time
    sd = orc.load_weights(os.path.join(ROOT, "bnv_fusion_amd", "weights", "pointnet_fp32.npz"))
    sel = pts.reshape(-1, 3)[torch.randperm(n_q, generator=torch.Generator().manual_seed(1))[:6000].to(dev)]
    cv = (sel - vol.min_coords) / voxel
    fl, ce = torch.floor(cv).long(), torch.ceil(cv).long()
    corners = torch.stack([torch.stack([(ce if b & 1 else fl)[:, 0], (ce if b & 2 else fl)[:, 1],
                                        (ce if b & 4 else fl)[:, 2]], -1) for b in range(8)], 1).reshape(-1, 3)
    keys = torch.unique(corners, dim=0)
    # the sample's corner voxels in the snapshot the decode reads (to_tensor(): the optimiser's count_optim has bumped
    # ITS weights, not the live table's)
    ac = vol.active_coordinates
    pack = lambda c: (c[:, 0] * 4096 + c[:, 1]) * 4096 + c[:, 2]      # noqa: E731
    order = torch.argsort(pack(ac))
    pos = torch.searchsorted(pack(ac)[order], pack(keys)).clamp(max=len(ac) - 1)
    rows = order[pos]
    present = (ac[rows] == keys).all(1)
    rows = rows[present]
    ovol = orc.OracleSparseVolume(8, voxel, np.asarray(vol.dimensions), 8)
    ovol.insert(keys[present].cpu(), vol.features[rows].detach().cpu(), vol.weights[rows].cpu(), vol.num_hits[rows].cpu())
    ovol.to_tensor()
    ovol.features.requires_grad_(True)
    q = sel.cpu().reshape(1, -1, 1, 3)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    t1 = time.perf_counter()
    ref = ovol.decode_pts(q, sd, None, is_coords=False, query_tensor=True)
    ref.sum().backward()
    t_cpu = time.perf_counter() - t1
    vol.features.grad = None
    got = vol.decode_pts(sel.reshape(1, -1, 1, 3), model.nerf, None)
    got.sum().backward()
    g_gpu = vol.features.grad[rows].cpu()
    g_ref = ovol.features.grad
    err_f = float((got.detach().cpu().reshape(-1) - ref.detach().reshape(-1)).abs().max())
    row_err = (g_gpu - g_ref).abs().amax(-1) / max(float(g_ref.abs().max()), 1e-30)
    err_g = float(row_err.max())
    rows_off = int((row_err > 1e-5).sum())
    vol.features = vol.features.detach()
    live_s = float((ref.detach() != voxel).float().mean())
    return {"what": "NeuralMap.optimize (run_e2e.py:111-162): Adam steps on the volume features; 5,000 rays of a random "
                    "key frame per step in 5 splits of 1,000 rays x (20 fine + 15 coarse) samples; all splits of a step in "
                    "one forward + loss + backward launch (bnv_optim_step), mask decisions and count_optim those of the "
                    "split-by-split sequence",
            "value": n_iters / dt, "unit": "optimisation steps/s", "ms_per_step": 1e3 * dt / n_iters, "steps": n_iters,
            "volume_rows": int(vol.num_rows()), "loss_first": float(hist[0]), "loss_last": float(hist[-1]),
            "step_launch_set": {"what": "optimize.ray_batch_step on a fixed batch of 5,000 rays: k_ray_samples + "
                                        "k_count_optim_splits + k_optim_step (forward + L1 loss + backward of all 5 splits) "
                                        "+ k_apply_split_counts + 2 uniform draws",
                                "avg_ms": step_ms, "queries": int(pred5.numel()), "live_queries_first_call": live5,
                                "algorithmic_tflops": 3 * live5 * 8.0 * FLOP_PER_EVAL / (step_ms * 1e-3) / 1e12,
                                "frac_of_peak": 3 * live5 * 8.0 * FLOP_PER_EVAL / (step_ms * 1e-3) / 1e12 / PEAK_TFLOPS[mlp_mode],
                                "algorithmic_flop": "live queries x 8 corners x 402,432 x 3 (forward + 2 x backward)"},
            "split": {"queries": n_q, "live_queries": live,
                      "k_decode_pts": {"avg_ms": fwd_ms, "tflops": flop_fwd / (fwd_ms * 1e-3) / 1e12,
                                       "frac_of_peak": flop_fwd / (fwd_ms * 1e-3) / 1e12 / peak,
                                       "algorithmic_flop": "live queries x 8 corners x 402,432"},
                      "k_decode_pts_bwd": {"avg_ms": bwd_ms, "tflops": 2 * flop_fwd / (bwd_ms * 1e-3) / 1e12,
                                           "frac_of_peak": 2 * flop_fwd / (bwd_ms * 1e-3) / 1e12 / peak,
                                           "algorithmic_flop": "2 x forward (recomputed forward + transposed chain)",
                                           "timing": "forward + backward minus forward (HIP events, 20 calls each)"},
                      "peak_tflops": peak},
            "parity": {"queries_checked": int(q.shape[1]), "live_fraction_checked": live_s,
                       "sdf_max_abs_err_vs_oracle": err_f, "tolerance": 1e-4,
                       "grad_max_err_over_max_grad_vs_oracle_autograd": err_g, "grad_tolerance": 1e-3,
                       "grad_rows_above_1e-5": rows_off, "grad_rows_checked": int(row_err.numel()),
                       "grad_note": "typically ~3e-7; the gradient of a ReLU network is discontinuous, and a "
                                    "pre-activation within rounding of zero (features after 43 Adam steps whose atomics "
                                    "sum in no fixed order) takes one path in one arithmetic and the other in the other: "
                                    "a run in five shows one or two rows at ~1e-4 (tests/test_gpu_optimize.py holds fixed "
                                    "inputs to 1e-4, and the reference's own loss gradient to 1e-3)"},
            "cpu_baseline": {"value": 1.0 / (t_cpu * n_q / q.shape[1] * 5), "unit": "optimisation steps/s", "kind": "port",
                             "cores": min(32, os.cpu_count() or 1),
                             "sample": f"oracle decode_pts forward + autograd backward of {q.shape[1]} queries: {t_cpu:.2f} s, "
                                       f"scaled to a step's 5 x {n_q} queries (ray sampling and Adam not counted)"}}


def extract_mesh_entry(nm, model, mlp_mode, label):
    """SURVEY.md section 8 f-4, NeuralMap.extract_mesh (run_e2e.py:164-167 -> SparseVolume.meshlize,
    sparse_volume.py:697-766) over a WHOLE volume: lattice decode of every active voxel + per-voxel marching cubes."""
    from bnv_fusion_amd import _lib, mesh as mesh_mod
    from oracle import bnv_oracle as orc           # checker / baseline only
    lib = _lib.load()
    vol = nm.volume
    dev = vol._dev
    voxel = vol.voxel_size
    delta = nm.prepare_tsdf_volume() if nm.tsdf_vol is not None else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = nm.extract_mesh()
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    m = nm.extract_mesh()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    vol.to_tensor()
    n_vox = int(vol.active_coordinates.shape[0])
    lib.bnv_profile_enable(1)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    sdf = vol.decode_lattice(vol.active_coordinates, model.nerf, delta, query_tensor=True)
    e1.record()
    verts, faces, _, _ = mesh_mod.marching_cubes_lattice_indexed(sdf, vol.active_coordinates, voxel, vol.min_coords)
    e2.record()
    torch.cuda.synchronize()
    ms, cnt = (C.c_double * 4)(), (C.c_int64 * 4)()
    lib.bnv_profile_read(ms, cnt)
    lib.bnv_profile_enable(0)
    evals = int(vol.last_lattice_evals()[0])
    tab_ms = ms[1] / max(cnt[1], 1)
    peak = PEAK_TFLOPS[mlp_mode]
    # parity + CPU time: the oracle's decode + marching cubes of a sample of voxels (with their neighbourhoods)
    sel = torch.randperm(n_vox, generator=torch.Generator().manual_seed(2))[:1024].to(dev)
    pick = vol.active_coordinates[sel]
    par = oracle_lattice_check(vol, vol.active_coordinates, sdf, n_voxels=1024, seed=2) if delta is None else None
    sd = orc.load_weights(os.path.join(ROOT, "bnv_fusion_amd", "weights", "pointnet_fp32.npz"))
    t1 = time.perf_counter()
    v_ref, f_ref = orc.meshlize_concat(sdf[sel].cpu().numpy().reshape(-1, 3, 3, 3), pick.cpu().numpy(), voxel,
                                       vol.min_coords.cpu().numpy())
    t_mc_cpu = time.perf_counter() - t1
    v_gpu, f_gpu, _, _ = mesh_mod.marching_cubes_lattice_indexed(sdf[sel], pick, voxel, vol.min_coords)
    same = (tuple(v_ref.shape) == tuple(v_gpu.shape) and tuple(f_ref.shape) == tuple(f_gpu.shape)
            and bool(np.abs(v_gpu.cpu().numpy() - v_ref).max() <= 1e-6 if len(v_ref) else True)
            and bool(np.array_equal(f_gpu.cpu().numpy(), f_ref)))
    cpu = cpu_decode_time(vol, pick[:256], sd) * n_vox / 256 + t_mc_cpu * n_vox / 1024
    return {"what": f"NeuralMap.extract_mesh over the whole {label} (run_e2e.py:164-167; sparse_volume.py:697-766): "
                    "to_tensor + lattice decode [M, 27] of every active voxel" + (" with the TSDF prior" if delta is not None else "")
                    + " + per-voxel marching cubes, vertices / faces to the host",
            "value": 1e3 * t_all, "unit": "ms per extract_mesh", "higher_is_better": False, "first_call_ms": 1e3 * t_first,
            "active_voxels": n_vox, "vertices": int(len(m.vertices)) if m is not None else 0,
            "faces": int(len(m.faces)) if m is not None else 0,
            "decode_ms": e0.elapsed_time(e1), "marching_cubes_ms": e1.elapsed_time(e2),
            "table_kernel": {"name": DECODE_KERNEL[mlp_mode], "avg_ms": tab_ms, "mlp_evals": evals,
                             "tflops": evals * FLOP_PER_EVAL / (tab_ms * 1e-3) / 1e12 if tab_ms else None,
                             "frac_of_peak": evals * FLOP_PER_EVAL / (tab_ms * 1e-3) / 1e12 / peak if tab_ms else None},
            "marching_cubes": {"bound": "hbm", "algorithmic_bytes": n_vox * 216 + int(len(verts)) * 12 + int(len(faces)) * 24,
                               "gb_per_s": (n_vox * 216 + int(len(verts)) * 12 + int(len(faces)) * 24) / (e1.elapsed_time(e2) * 1e-3) / 1e9,
                               "note": "108 B of lattice per voxel read by the count pass and again by the emit pass, "
                                       "12 B per vertex + 24 B per triangle (int64 indices) written"},
            "parity": {"sdf": par, "marching_cubes_equal_oracle_on_1024_voxels": same,
                       "note": "triangulation of ambiguous cells follows the oracle's restatement; scikit-image's "
                               "Lewiner tables are absent from this image (parity unpinned, DESIGN.md section 4)"},
            "cpu_baseline": {"value": 1e3 * cpu, "unit": "ms per extract_mesh", "kind": "port", "cores": min(32, os.cpu_count() or 1),
                             "sample": "oracle decode of 256 voxels' lattices + oracle marching cubes of 1,024 voxels, "
                                       "scaled to the volume's active voxels"}}


def cpu_decode_time(volume, pick, sd):
    """Seconds the oracle needs for the lattice decode of ``pick`` (their neighbourhoods read back from the GPU)."""
    from oracle import bnv_oracle as orc           # baseline only
    dev = pick.device
    off = torch.tensor([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)], device=dev)
    nbr = torch.unique((pick[:, None, :] + off[None]).reshape(-1, 3), dim=0)
    fo, wo, _ = volume.query(nbr)
    ovol = orc.OracleSparseVolume(8, volume.voxel_size, np.asarray(volume.dimensions), 8)
    present = wo[:, 0].cpu() > 0
    ovol.insert(nbr.cpu()[present], fo.cpu()[present], wo.cpu()[present], torch.zeros(int(present.sum()), 1))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        t0 = time.perf_counter()
        ovol.decode_pts(orc.lattice_coords(pick.cpu().numpy()), sd, None, is_coords=True, query_tensor=False)
        return time.perf_counter() - t0


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (a fresh
    process tree; this process has not touched the GPU and never will) and leave with the child's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this image
    env.setdefault("OMP_NUM_THREADS", "4")
    env["BNV_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.run(cmd, env=env).returncode


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=256, choices=[128, 256, 512])
    ap.add_argument("--preroll", type=int, default=30,
                    help="frames fused (untimed setup) before warm-up so that voxel weights reach "
                         "min_pts_in_grid and the decode mask is live (SURVEY.md section 8d)")
    ap.add_argument("--preheat", type=int, default=1000,
                    help="untimed frames run back to back immediately before the warm-up + timed steps, so that the "
                         "clock has settled under the package power limit when the timed region starts (`value` is a "
                         "sustained rate; the rate from an idle GPU is reported as `burst`); 0 = none.  1,000 frames "
                         "(~1.8 s): behind 320 the timed steps still ran 3.5 % above the 1,000-frame `sustained` pass, "
                         "behind 1,000 within 1.1 %")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-widened", action="store_true",
                    help="skip the entries of the widened rows (`optimize`, `extract_mesh`, `extract_mesh_sweep`)")
    ap.add_argument("--mlp-mode", type=int, default=1, choices=[0, 1, 3],
                    help="1 (default): split-f16 operands on the f16 MFMA; 0: exact fp32 MFMA")
    ap.add_argument("--no-alt-mode", action="store_true",
                    help="skip the fp32_exact run, the f16-operand run, the sustained pass, the growth run, the long "
                         "sequence and (N > 1) the second decomposition")
    ap.add_argument("--no-power-probe", action="store_true",
                    help="skip the MFMA-only rate probe (roofline.power_limited_mfma_ceiling)")
    ap.add_argument("--sustained-frames", type=int, default=1000,
                    help="frames of the sustained pass (1 GPU; 0 = skip)")
    ap.add_argument("--sequence-frames", type=int, default=600,
                    help="frames of the moving-camera sequence pass (1 GPU; 0 = skip): the room-sweep trajectory of "
                         "bnv_fusion_amd.synthetic.sweep_pose over a 512^3 volume that grows from the reference's "
                         "initial capacity while two frames are in flight")
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
    ap.add_argument("--parallelism", default="spatial", choices=["frame", "spatial"],
                    help="N > 1: which decomposition `value` reports (the other one is measured in the same run and "
                         "reported beside it).  'spatial' (default; the decomposition BASELINE.json names): the "
                         "active-voxel set sharded by spatial hash, one RCCL all-gather of boundary-voxel records per "
                         "frame, every frame worked on by all ranks (strong scaling of one stream); 'frame': ranks "
                         "encode/decode different frames of a batch, replicated volume, one all-gather per batch "
                         "(weak scaling)")
    ap.add_argument("--dist-timeout", type=float, default=180.0,
                    help="N > 1: seconds after which a collective (or the rendezvous) gives up instead of hanging")
    ap.add_argument("--first-contact-timeout", type=float, default=60.0,
                    help="N > 1: seconds each of the five warm-up all-gathers in front of the timed region may take "
                         "(bnv_fusion_amd.distributed.first_contact: device identities, one GPU per rank, data checked)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="launcher test (no GPU needed): every rank checks RANK / WORLD_SIZE against --gpus, rank 0 "
                         "prints them, all exit before any GPU call")
    return ap.parse_args()


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if args.dry_run_launch:
        if rank == 0:
            print(json.dumps({"dry_run_launch": True, "world": world, "gpus": args.gpus,
                              "self_launched": os.environ.get("BNV_BENCH_SELF_LAUNCHED") == "1"}))
        return
    # one rank per GPU; BNV_DIST_BACKEND=gloo lets several ranks share a GPU for functional testing
    backend = os.environ.get("BNV_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()              # (does not initialise the GPU)
    if backend == "nccl" and world > 1 and local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPUs visible -- RCCL needs one GPU "
                         "per rank (BNV_DIST_BACKEND=gloo lets ranks share a GPU for functional tests)")
    local_dev = local_rank % max(n_dev, 1)
    torch.cuda.set_device(local_dev)
    dev = f"cuda:{local_dev}"
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        # a wedged rank must fail the run, not hang it: every collective (and the rendezvous) times out
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        if backend == "nccl":
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
            # RCCL's stream at high priority: the exchange kernels are short and on the path of the next batch;
            # they should take a CU as soon as one of the (CU-filling) compute kernels releases it
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", device_id=torch.device(dev), pg_options=opts, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    try:
        out = run_bench(args, rank, world, dev, dist, backend)
    except BaseException:
        # any rank's failure ends the job with a non-zero code (torch.distributed.run then stops the other ranks,
        # and the self-launching parent leaves with the child's code)
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


class Driver:
    """Runs frames through a map of one of three kinds: 'single' (NeuralMap) and 'spatial' (ShardedNeuralMap) hand
    out one handle per frame; 'frame' (FrameParallelNeuralMap) works on batches of `world` frames."""

    def __init__(self, args, frames, world):
        self.args, self.frames, self.world = args, frames, world

    def run(self, m, kind, idx, decode=True, collect=None):
        """Processes the frames with indices ``idx`` in order on map ``m``; returns this rank's last (coords, sdf)."""
        frames, world = self.frames, self.world
        last = (None, None)
        if kind == "frame":
            # batches of `world` consecutive frames, software-pipelined (batch k+1's encode + all-gather are
            # enqueued before batch k's integrate + decode); only the last handle is read back on the host
            batches = [[frames[t] for t in idx[b0: b0 + world]] for b0 in range(0, len(idx), world)]
            handle = None
            for handle in m.process_stream(batches, decode=decode):
                pass
            m.flush()
            if handle is not None:
                out = handle.result()
                if out[0] is not None:
                    last = out
        elif not self.args.sync_frames:
            # software pipeline: frame t is enqueued before frame t-1's result is collected, so the GPU never waits
            # for the host (each result() waits on that frame's own event only).  Frames enqueued ahead of the oldest
            # uncollected one -- 1 GPU: 3 (2 is +5 % against 1; on the four-stream pipeline 3 is neutral for the fp32
            # checkpoint, 546-551 frames/s either way on one box, and +0-2 % / +3-4 % burst with the tiny-cuda-nn
            # networks, whose frame is short against the host's enqueue time); sharded: 3 (the
            # encode stream then runs a whole frame ahead of the main stream: 0.279 against 0.315 ms per frame at a
            # simulated world of 8)
            from collections import deque
            pending = deque()
            depth = int(os.environ.get("BNV_BENCH_DEPTH", "3"))
            _dbg = [] if os.environ.get("BNV_BENCH_DEBUG") else None
            for j, t in enumerate(idx):
                _a = time.perf_counter()
                if kind == "spatial":   # the next frame's encode is enqueued before the host waits for this one's bound
                    nxt = frames[idx[j + 1]] if j + 1 < len(idx) else None
                    h = m.fuse_and_decode_async(frames[t], decode=decode, next_frame=nxt)
                else:
                    h = m.fuse_and_decode_async(frames[t], decode=decode)
                if _dbg is not None:
                    _dbg.append(time.perf_counter() - _a)
                pending.append(h)
                if len(pending) > depth:
                    r = pending.popleft().result()
                    if collect is not None:
                        collect(r)
            while pending:
                last = pending.popleft().result()
                if collect is not None:
                    collect(last)
            if _dbg:
                print("enqueue ms:", " ".join(f"{1e3*x:.2f}" for x in _dbg), file=sys.stderr)
        else:
            for t in idx:
                last = m.fuse_and_decode(frames[t]) if decode else (m.integrate(frames[t]), None)
                if collect is not None:
                    collect(last)
        return last


def run_bench(args, rank, world, dev, dist, backend):
    import bnv_fusion_amd as bnv
    from bnv_fusion_amd import synthetic, _lib

    dims, voxel = synthetic.GRID_DIMS[args.grid]
    dims3 = np.array([dims] * 3)
    tcnn = args.checkpoint == "tcnn"
    model = bnv.load_pretrained(device=dev, voxel_size=voxel, tiny_cuda=tcnn)
    if tcnn:
        args.mlp_mode = 2
    with_tsdf = args.input == "depth"
    lib = _lib.load()

    def make_map(kind):
        if kind == "frame":
            from bnv_fusion_amd.distributed import FrameParallelNeuralMap
            m = FrameParallelNeuralMap(dims3, voxel, model, device=dev, tsdf=with_tsdf, capacity=CAPACITY)
            m.backend.inputs_resident = True
        elif kind == "spatial":
            from bnv_fusion_amd.distributed import ShardedNeuralMap
            m = ShardedNeuralMap(dims3, voxel, model, device=dev, capacity=CAPACITY, tsdf=with_tsdf)
            m.backend.inputs_resident = True      # every frame is uploaded (and synchronised) before anything is timed
            m.backend.copy_results = False        # results are views into the slot ring (collected before reuse)
            m.backend.n_slots = 5
        else:
            m = bnv.NeuralMap(dims3, voxel, model, capacity=CAPACITY, device=dev, tsdf=with_tsdf)
            m.overlap_encode = not args.no_stream_overlap
            m.inputs_resident = True
            m.copy_results = False            # results are views into the pipeline's slots (collected before reuse)
        return m

    kinds = ["single"] if world == 1 else ([args.parallelism] + ([] if args.no_alt_mode else
                                           [k for k in ("spatial", "frame") if k != args.parallelism]))

    # ---- synthetic inputs, resident in HBM before anything is timed -----------------------------
    # frames per step: one frame on one GPU and in the sharded mode (every rank works on every frame); in
    # frame-parallel mode a step is one batch = one frame PER RANK (weak scaling, `value` counts all of them)
    POOL = 64                      # two pan periods (synthetic.yaw_deg): long passes cycle over them
    fpu_max = world if "frame" in kinds else 1
    n_frames = args.preroll + max((args.warmup + args.steps) * fpu_max, POOL)
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
    torch.cuda.synchronize()
    drv = Driver(args, frames, world)
    pool = list(range(args.preroll, args.preroll + POOL))      # 2 pan periods: cycling continues the pan

    def timed(m, kind, mode, idx, warm_idx, preheat=0):
        """[`preheat` untimed frames of the pool, back to back,] untimed warm-up over ``warm_idx``, then times exactly
        the steps ``idx`` (a multiple of the frames per step), in MLP mode ``mode``.  Nothing happens on the host
        between the pre-heat and the timed steps but a stream synchronisation."""
        fpu = world if kind == "frame" else 1
        if mode != 2:
            bnv.set_mlp_mode(mode)
        # the cyclic garbage collector stays out of the timed region (as timeit does): with torch imported a full
        # collection takes ~40 ms, and one landed on the third timed frame of every process but the first on a box
        # (350 instead of 540 frames/s over 40 frames; found with per-frame enqueue times, BNV_BENCH_DEBUG=1)
        gc.collect()
        gc.disable()
        if preheat:
            pre = [pool[i % POOL] for i in range(-(-preheat // fpu) * fpu)]
            drv.run(m, kind, pre)
        drv.run(m, kind, warm_idx)
        lib.bnv_profile_enable(1)
        n_vox = []
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()

        n_pairs, n_evals = [], []

        def collect(res):
            c, _ = res
            n_vox.append(0 if c is None else int(c.shape[0]))
            if kind == "spatial":        # this rank's share of the frame (bnv_encode_counters_t.reserved[0], pinned words)
                n_pairs.append(int(m.backend.last_owned_pairs))
                n_evals.append(int(m.backend._last_evals))

        coords, sdf = drv.run(m, kind, idx, collect=None if kind == "frame" else collect)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        gc.enable()
        if kind == "frame":
            n_vox.append(0 if coords is None else int(coords.shape[0]))
        evals = float(m.volume.last_lattice_evals().item()) if coords is not None else 0.0
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        prof_ms = (C.c_double * 4)()
        prof_n = (C.c_int64 * 4)()
        lib.bnv_profile_read(prof_ms, prof_n)
        lib.bnv_profile_enable(0)
        live = float((sdf != voxel).float().mean()) if sdf is not None and sdf.numel() else 0.0
        dec_ms = prof_ms[1] / max(prof_n[1], 1)
        enc_ms = prof_ms[0] / max(prof_n[0], 1)
        dec_flop = evals * (FLOP_PER_EVAL_TCNN if mode == 2 else FLOP_PER_EVAL)
        share = world if kind == "spatial" else 1          # a rank encodes 1 / world of the pairs when sharded
        enc_flop = 8.0 * n_points * (FLOP_PER_PAIR_TCNN if mode == 2 else FLOP_PER_PAIR) / share
        steps = len(idx) // fpu
        return {"elapsed": elapsed, "steps": steps, "fps": len(idx) / elapsed, "rows": evals,
                "n_vox": float(np.mean(n_vox)) if n_vox else 0.0, "live": live, "dec_ms": dec_ms, "enc_ms": enc_ms,
                "dec_tflops": dec_flop / (dec_ms * 1e-3) / 1e12 if dec_ms else 0.0,
                "enc_tflops": enc_flop / (enc_ms * 1e-3) / 1e12 if enc_ms else 0.0, "dec_flop": dec_flop,
                "coords": coords, "sdf": sdf, "kind": kind, "fpu": fpu,
                "pairs_per_frame": float(np.mean(n_pairs)) if n_pairs else 0.0,
                "evals_per_frame": float(np.mean(n_evals)) if n_evals else 0.0,
                "frames_this_rank": len(idx) // world if kind == "frame" else len(idx)}

    # parity check of a configuration against the oracle (PARITY_VOXELS voxels of its last frame): the SDF lattice
    # decoded by the GPU from the GPU's own volume vs the oracle's decode of the same volume values.  Sharded: rank 0
    # checks voxels IT owns -- their neighbourhoods are its own rows + the ghost rows the exchange installed
    def parity_check(m, run):
        if not (rank == 0 and run["coords"] is not None):
            return None
        # frame-parallel: the replicated volume has moved on behind this rank's frame: decode again from its current state
        again = (lambda pick: m.volume.decode_lattice(pick, model.nerf, query_tensor=False)) if run["kind"] == "frame" \
            else None
        return oracle_lattice_check(m.volume, run["coords"], run["sdf"], tcnn=tcnn, decode_again=again)

    def idx_of(fpu):
        first = args.preroll + args.warmup * fpu
        return list(range(args.preroll, first)), list(range(first, first + args.steps * fpu))

    def roofline_of(mode, kern, note):
        peak = PEAK_TFLOPS[mode]
        traffic, src = (None, None)
        if mode == 1 and world == 1 and not tcnn:
            traffic, src = pmc_traffic("k_lattice_table_x<3>", kern["rows"])
        # achieved = algorithmic FLOPs (402,432 per MLP evaluation x evaluations per launch; the split mode issues 3
        # MFMA products per algorithmic product, which are NOT counted) / mean kernel time from HIP events recorded
        # on the launch stream
        return {"bound": "mfma", "kernel": DECODE_KERNEL[mode] + (" (SDF MLP 32-64x3-16, tiny-cuda-nn layout)" if mode == 2
                                                                  else " (SDF MLP 17-256x4-1)"),
                "achieved": kern["dec_tflops"], "peak": peak, "unit": "TFLOP/s", "frac": kern["dec_tflops"] / peak,
                "traffic": traffic, "traffic_source": src,
                "avg_kernel_ms": kern["dec_ms"], "flop_per_launch": kern["dec_flop"],
                "mlp_evals_per_launch": kern["rows"],
                "mfma_issue_frac": kern["dec_tflops"] * MFMA_PER_PRODUCT[mode] / peak, "timing": note,
                "state": "sustained: measured behind the pre-heat frames like `value` (the same kernel runs ~20 % "
                         "faster on a GPU that has been idle: burst.kernel_alone)" if world == 1 else "timed region"}

    def entry(run, parity=None):
        return {"value": run["fps"], "unit": "frames/s", "steps": run["steps"],
                "ms_per_step": 1e3 * run["elapsed"] / run["steps"], **({"parity": parity} if parity else {}),
                **({"kernel_alone": run["kernel_alone"]} if "kernel_alone" in run else {})}

    out_common = {"metric": "depth frames/sec fused+decoded, 640x480 @ 256^3 grid", "unit": "frames/s",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                  "vs_baseline": None, "dtype": DTYPE[args.mlp_mode], "data": "synthetic"}
    workload = (f"synthetic 640x480 depth ({n_points} valid points/frame; the analytic surface of BASELINE.md section 3 as a "
                f"STATIC scene seen by a camera panning +-4 degrees in 0.5-degree steps -- BASELINE.md glues the pattern to "
                f"the camera, which leaves no voxel with weight >= 8 and a 93 % masked decode), {args.grid}^3 grid, voxel {voxel}, "
                f"{'pointnet_tcnn.ckpt (fp16 tcnn)' if tcnn else 'fp32 pointnet.ckpt'} weights; step = "
                + ("uint16 depth image -> points + normals (GPU front end) + " if args.input == "depth" else "")
                + "encode_pointcloud + _integrate + "
                + ("TSDF side fusion at 0.025 m + " if args.input == "depth" else "")
                + "decode of the 3x3x3 lattice of every touched voxel")

    # =========================================================================================================
    # N > 1: both decompositions, the one --parallelism names in `value`
    # =========================================================================================================
    if world > 1:
        # first contact (before anything is timed): who is there, one distinct GPU per rank, five all-gathers shaped
        # like a frame's exchange with their own timeout and their data checked -- a failure raises, main() exits
        # non-zero, the launcher stops the other ranks
        from bnv_fusion_amd.distributed import first_contact
        contact = first_contact(rank, world, dev, backend=backend, timeout_s=args.first_contact_timeout,
                                identity=(None if backend == "nccl" else (socket.gethostname(), os.getpid())))
        results, errors = {}, {}
        for kind in kinds:
            # A decomposition that raises on this rank (an argument the backend rejects, a capacity, ...) is reported
            # and the other one still measured -- every rank votes, so that all of them skip it together.  A rank that
            # hangs or dies is not survivable: the collective times out and the job ends non-zero (main()).
            m, run, extra, err = None, None, {}, None
            try:
                if os.environ.get("BNV_BENCH_FAIL_KIND") == kind:        # (fault injection for the tests)
                    raise RuntimeError("injected failure")
                m = make_map(kind)
                fpu = world if kind == "frame" else 1
                drv.run(m, kind, list(range(-(-args.preroll // fpu) * fpu)), decode=False)     # setup: live decode mask
                warm_idx, step_idx = idx_of(fpu)
                run = timed(m, kind, args.mlp_mode, step_idx, warm_idx, preheat=min(args.preheat, 4 * POOL))
                run["parity"] = parity_check(m, run)
                if kind == "spatial":
                    extra = {"ownership": m.backend.ownership, "block_log2": m.backend.block_log2,
                             "exchange": "rows after the upsert, one all-gather on the main stream",
                             "received_bytes_per_frame_and_rank": m.exchanged_bytes / max(m.host_waits, 1),
                             "host_waits_per_frame": 1, "encode_stream_overlaps_main_stream":
                                 bool(getattr(m.backend.pipe.enc, "bnv_concurrent", False))}
            except Exception as e:       # noqa: BLE001 -- reported in the output line
                import traceback
                traceback.print_exc()
                err = f"rank {rank}: {type(e).__name__}: {e}"
            votes = [None] * world
            dist.all_gather_object(votes, {"error": err, "frames": int(run["frames_this_rank"]) if run else 0,
                                           "voxels_per_frame": run["n_vox"] if run else 0,
                                           "pairs_per_frame": run["pairs_per_frame"] if run else 0,
                                           "mlp_evals_per_frame": run["evals_per_frame"] if run else 0,
                                           "mlp_evals_last_frame": run["rows"] if run else 0,
                                           "ms_per_frame": 1e3 * run["elapsed"] / max(run["steps"], 1) if run else 0})
            failed = [v["error"] for v in votes if v["error"]]
            if failed:
                errors[kind] = failed
            else:
                keys = ("frames", "voxels_per_frame", "pairs_per_frame", "mlp_evals_per_frame", "mlp_evals_last_frame")
                extra["per_rank"] = [{k: v[k] for k in keys} for v in votes]
                if kind == "spatial":       # what the rank set ran with: the slowest rank sets the pace
                    def mom(k):
                        a = np.array([float(v[k]) for v in votes])
                        return float(a.max() / a.mean()) if a.mean() > 0 else None
                    extra["load_max_over_mean"] = {k: mom(k) for k in ("voxels_per_frame", "pairs_per_frame",
                                                                       "mlp_evals_per_frame")}
                results[kind] = (run, extra)
            del m
            model.shard = (0, 1, 3)
            torch.cuda.empty_cache()
        if not results:
            raise RuntimeError(f"every decomposition failed: {errors}")
        kinds = [k for k in kinds if k in results]       # (the first one that ran is `value`)
        if rank != 0:
            return None
        from bnv_fusion_amd.distributed import DEFAULT_OWNERSHIP
        own_rule = results["spatial"][1].get("ownership", DEFAULT_OWNERSHIP) if "spatial" in results else DEFAULT_OWNERSHIP
        rule = {"region": "first-touch block ownership in contiguous regions (8^3-voxel blocks; the first frame is cut "
                          "into bands of equal load, later blocks go to the owner of a neighbour block unless it carries "
                          "more than its share; every rank derives the same table, no communication)",
                "first_touch": "first-touch block ownership (8^3-voxel blocks, a new block goes to the least-loaded rank; "
                               "every rank derives the same table, no communication)",
                "hash": "spatial hash of 8^3-voxel blocks"}[own_rule]
        DESCR = {"spatial": (f"active-voxel set sharded over {world} ranks by {rule}; every "
                             "rank voxelises the whole frame, encodes / upserts / decodes the voxels it owns; per frame "
                             "ONE RCCL all-gather of boundary-voxel records (48 B: key, weight, 8 features) and one host "
                             "wait (the exchange bound, read while the previous frame decodes); a step = one frame "
                             "worked on by all ranks", "strong"),
                 "frame": (f"frame-parallel x{world}: a step = one batch of {world} consecutive frames; ranks encode / "
                           "decode different frames of the batch, replicated volume, one RCCL all-gather of encoded "
                           "voxels per batch", "weak")}

        def dist_entry(kind):
            run, extra = results[kind]
            return {**entry(run, run["parity"]), "scaling": DESCR[kind][1], "decomposition": DESCR[kind][0],
                    "frames_per_step": run["fpu"], "voxels_per_frame_rank0": run["n_vox"],
                    "decode_live_fraction_rank0": run["live"], "decode_kernel_ms_rank0": run["dec_ms"],
                    "pointnet_kernel_ms_rank0": run["enc_ms"], **extra}

        prim, _ = results[kinds[0]]
        m_ = args.mlp_mode
        out = dict(out_common)
        out.update({
            "value": prim["fps"], "ms_per_step": 1e3 * prim["elapsed"] / prim["steps"], "scaling": DESCR[kinds[0]][1],
            "config": {"workload": workload, "grid": args.grid, "voxel_size": voxel, "preroll_frames": args.preroll,
                       "preheat_frames": min(args.preheat, 4 * POOL), "frames_per_step": prim["fpu"],
                       "mlp_mode": MODE_NAME[m_], "parallelism": DESCR[kinds[0]][0]},
            "roofline": roofline_of(m_, prim, "rank 0's launches inside the timed region (the two streams of the frame "
                                              "pipeline overlap: durations include time shared with the other "
                                              "stream's kernels)"),
            "kernels": {"pointnet_scatter": {"avg_ms": prim["enc_ms"], "tflops": prim["enc_tflops"],
                                             "frac_of_peak": prim["enc_tflops"] / PEAK_TFLOPS[m_]}},
            "parity": prim["parity"],
            "distributed": {"backend": "rccl" if backend == "nccl" else backend,
                            "world_size_seen_by_backend": dist.get_world_size(), **contact,
                            "collective_timeout_s": args.dist_timeout,
                            "launcher": ("bench.py started torch.distributed.run itself"
                                         if os.environ.get("BNV_BENCH_SELF_LAUNCHED") == "1"
                                         else "external torch.distributed.run")},
            "spatial_sharding" if kinds[0] == "spatial" else "frame_parallel": dist_entry(kinds[0]),
        })
        for kind in kinds[1:]:
            out["spatial_sharding" if kind == "spatial" else "frame_parallel"] = dist_entry(kind)
        for kind, errs in errors.items():
            out["spatial_sharding" if kind == "spatial" else "frame_parallel"] = {"error": errs}
        return out

    # =========================================================================================================
    # one GPU
    # =========================================================================================================
    nm = make_map("single")
    drv.run(nm, "single", list(range(args.preroll)), decode=False)           # setup: make the decode mask live
    warm_idx, step_idx = idx_of(1)

    def measure(mode):
        """-> (sustained run: pre-heated timed region [+ parity], burst run: the same steps from an idle GPU,
        kernel-alone run, note)."""
        ph = args.preheat if mode != 0 else min(args.preheat, 300)     # (exact fp32 draws less: burst == sustained)
        burst = timed(nm, "single", mode, step_idx, warm_idx)
        if getattr(nm, "overlap_encode", False):      # the dominant kernel alone at the burst clock, as rounds 1-2 quoted it
            nm.overlap_encode = False
            kb = timed(nm, "single", mode, step_idx, warm_idx)
            nm.overlap_encode = True
            burst["kernel_alone"] = {"decode_kernel_ms": kb["dec_ms"], "decode_tflops": kb["dec_tflops"],
                                     "decode_frac_of_peak": kb["dec_tflops"] / PEAK_TFLOPS[mode],
                                     "pointnet_kernel_ms": kb["enc_ms"], "pointnet_tflops": kb["enc_tflops"]}
        run = timed(nm, "single", mode, step_idx, warm_idx, preheat=ph)
        run["parity"] = parity_check(nm, run)  # right after the timed region: the volume is in that run's final state
        # Kernel-alone pass for the roofline: with the encode on a second stream the two MLP kernels of consecutive
        # frames share the GPU, so their event-to-event durations overlap.  A roofline needs the kernel's own
        # duration: the same frames are run again with everything on one stream (and that is how the committed
        # rocprofv3 summaries are taken: --no-stream-overlap), pre-heated like the timed region.
        kern, note = run, "timed region"
        if getattr(nm, "overlap_encode", False):
            nm.overlap_encode = False
            kern = timed(nm, "single", mode, step_idx, warm_idx, preheat=ph)
            nm.overlap_encode = True
            note = (f"the same {kern['steps']} frames (+{args.warmup} warm-up, behind {ph} pre-heat frames) "
                    "re-run with the encode on the main stream, so that each kernel has the GPU to itself (in the timed "
                    f"region the two MLP kernels of consecutive frames overlap); that pass ran at {kern['fps']:.1f} "
                    "frames/s")
        return run, burst, kern, note

    main_run, burst_run, kern_run, kern_note = measure(args.mlp_mode)

    def sustainable_mfma():
        """What the f16 MFMA pipe of THIS box sustains when nothing else is issued (bnv_probe_mfma_rate: every CU, two
        waves per SIMD, ~8 ms per case): the MLP kernels run at the package power limit, where the clock settles below
        the 2.4 GHz the 2.5 PFLOP/s peak is quoted at -- by how much depends on the MFMA shape (the 16x16x32 form the
        dominant kernel uses moves half the accumulator data per FLOP) and on the operand data (zeros toggle nothing).
        Measured right behind the timed frames, GPU warm."""
        res = {}
        for name, shape, operands, iters in (("16x16x32_random_f16_operands", 1, 1, 16000),
                                             ("32x32x16_random_f16_operands", 0, 1, 16000),
                                             ("16x16x32_zero_operands", 1, 0, 8000)):
            ms, flop = C.c_double(), C.c_double()
            _lib.check(lib.bnv_probe_mfma_rate(shape, operands, iters,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                               C.byref(ms), C.byref(flop)), "bnv_probe_mfma_rate")
            res[name] = flop.value / (ms.value * 1e-3) / 1e12
        return res

    power_ceiling = None
    if not tcnn and args.mlp_mode in (1, 3) and not args.no_power_probe:
        pc = sustainable_mfma()
        issued = kern_run["dec_tflops"] * MFMA_PER_PRODUCT[args.mlp_mode]
        power_ceiling = {
            "what": "rate of an MFMA-ONLY stream on this GPU (every CU, two waves per SIMD, ~8 ms per case, right "
                    "behind the timed frames): the package power limit, not the 2.4 GHz clock ceiling behind "
                    "roofline.peak; the dominant kernel issues v_mfma_f32_16x16x32_f16",
            "tflops_16x16x32_random_f16_operands": pc["16x16x32_random_f16_operands"],
            "tflops_32x32x16_random_f16_operands": pc["32x32x16_random_f16_operands"],
            "tflops_16x16x32_zero_operands": pc["16x16x32_zero_operands"],
            "dominant_kernel_issued_tflops": issued,
            "dominant_kernel_frac_of_it": issued / pc["16x16x32_random_f16_operands"]}

    extras = {}
    alts = []
    pipe_ = getattr(nm, "_pipe", None)
    if pipe_ is not None and getattr(pipe_, "persistent_tables", False) and not args.no_alt_mode:
        # ---- A/B inside this run: the same timed region with every frame's table entries recomputed (the persistent
        # lattice tables off), so that what the carried-over entries are worth is a figure of THIS box
        nm._drain_pipe()
        pipe_.persistent_tables = False
        rn = timed(nm, "single", args.mlp_mode, step_idx, warm_idx,
                   preheat=args.preheat if args.mlp_mode != 0 else min(args.preheat, 300))
        nm._drain_pipe()
        pipe_.persistent_tables = True
        nm.volume.invalidate_tables()
        extras["without_persistent_tables"] = {**entry(rn), "mlp_evals_last_timed_frame": rn["rows"],
                                               "note": "every frame re-evaluates all of its table entries"}
    if not args.no_alt_mode and not tcnn:
        # ---- the reference's precision to the letter: IEEE fp32 MFMA, the full --steps, its own roofline -------
        if args.mlp_mode != 0:
            r0, b0, k0, n0 = measure(0)
            extras["fp32_exact"] = {
                "dtype": DTYPE[0], **entry(r0, r0["parity"]), "burst": entry(b0), "roofline": roofline_of(0, k0, n0),
                "kernels": {"pointnet_scatter": {"avg_ms": k0["enc_ms"], "tflops": k0["enc_tflops"],
                                                 "frac_of_peak": k0["enc_tflops"] / PEAK_TFLOPS[0]}}}
        # ---- lower precision, reported for completeness (never the headline) ------------------------------------
        for am in (1, 3):
            if am == args.mlp_mode:
                continue
            r = timed(nm, "single", am, step_idx[: 8], warm_idx[-3:])
            r["mode"], r["parity"] = am, parity_check(nm, r)
            alts.append(r)
        bnv.set_mlp_mode(args.mlp_mode)

    if not args.no_alt_mode:
        # ---- sustained pass: >= 1,000 frames back to back in the headline mode, clock + power sampled ------------
        if args.sustained_frames:
            idx = [pool[i % POOL] for i in range(args.sustained_frames)]
            drv.run(nm, "single", pool[:8])
            torch.cuda.synchronize()
            gc.collect()
            gc.disable()
            with SmiSampler() as smi:
                t0 = time.perf_counter()
                drv.run(nm, "single", idx)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            gc.enable()
            summ = smi.summary(t0, t1)
            extras["sustained"] = {"frames": len(idx), "value": len(idx) / (t1 - t0), "unit": "frames/s",
                                   "ms_per_frame": 1e3 * (t1 - t0) / len(idx), "mlp_mode": MODE_NAME[args.mlp_mode],
                                   "frames_note": f"the {POOL} frames after the pre-roll (two periods of the +-4 degree "
                                                  "pan), cycled; the volume keeps accumulating",
                                   **summ,
                                   "joules_per_frame": (summ["mean_package_power_w"] * (t1 - t0) / len(idx)
                                                        if summ.get("mean_package_power_w") else None)}

        # ---- the reference's DEFAULT networks (pointnet_tcnn.ckpt; parity unpinned), briefly: the
        # same timed steps on a volume of their own, so that the default line carries a figure for them too
        # (`--checkpoint tcnn` is the full run)
        if not tcnn:
            mt = bnv.load_pretrained(device=dev, voxel_size=voxel, tiny_cuda=True)
            nmt = bnv.NeuralMap(dims3, voxel, mt, capacity=CAPACITY, device=dev, tsdf=with_tsdf)
            nmt.inputs_resident, nmt.copy_results = True, False
            drv.run(nmt, "single", list(range(args.preroll)), decode=False)
            # (50 of these frames are 11 ms: one hiccup of the box halves the figure -- the median of three passes)
            rts = [timed(nmt, "single", 2, step_idx, warm_idx, preheat=min(args.preheat, 300) if k == 0 else 0)
                   for k in range(3)]
            rt = sorted(rts, key=lambda r: r["elapsed"])[1]
            chk = oracle_lattice_check(nmt.volume, rt["coords"], rt["sdf"], tcnn=True, n_voxels=256) \
                if rt["coords"] is not None else None
            pq = getattr(nmt, "_pipe", None)
            extras["tcnn_quick"] = {**entry(rt, chk), "dtype": DTYPE[2],
                                    "passes_frames_per_s": [r["fps"] for r in rts],
                                    "pipe_streams_concurrent": None if pq is None else [
                                        getattr(st, "bnv_concurrent", None) for st in (pq.enc, pq.front, pq.blend)],
                                    "note": "tiny-cuda-nn networks of the reference's default checkpoint, the same "
                                            "timed steps behind 300 pre-heat frames, the median of three passes; parity UNPINNED (the reference's "
                                            "fp16 arithmetic is CUDA-only): checked against the oracle's restatement"}
            del nmt, mt
            model.shard = (0, 1, 3)
            bnv.set_mlp_mode(args.mlp_mode)
            torch.cuda.empty_cache()

        # ---- growth: from an EMPTY volume of the reference's initial capacity (sparse_volume.py:486: 100,000) ---
        nm2 = bnv.NeuralMap(dims3, voxel, model, capacity=100000, device=dev, tsdf=with_tsdf)
        nm2.overlap_encode, nm2.inputs_resident = nm.overlap_encode, True
        cap0 = nm2.volume._row_capacity
        n_g = min(60, n_frames)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        drv.run(nm2, "single", list(range(n_g)))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        extras["growth"] = {"frames": n_g, "value": n_g / (t1 - t0), "unit": "frames/s",
                            "row_capacity_start": cap0, "row_capacity_end": nm2.volume._row_capacity,
                            "rows_end": nm2.volume.num_rows(),
                            "note": "fuse+decode from an empty volume that grows on demand (sync + re-allocation + "
                                    "re-hash per doubling); early frames decode to the masked constant (weights < "
                                    "min_pts), so this prices the growth steps, not the steady state"}
        del nm2

        # ---- a long moving-camera sequence: the surrogate of BASELINE configs 0 / 2 / 4 (datasets absent) ----------
        if args.sequence_frames and args.input == "depth":
            from bnv_fusion_amd import sequence
            def seq_check(nm_s, c, sdf_s):
                r = oracle_lattice_check(nm_s.volume, c, sdf_s, tcnn=tcnn, n_voxels=256)
                return r["sdf_max_abs_err_vs_oracle"], r["mask_decisions_equal"], r["live_fraction_checked"]

            kept = []
            extras["sequence"] = sequence.bench_pass(model, dev, args.sequence_frames, check=seq_check, keep=kept)
            if not args.no_widened and not tcnn and kept:
                bnv.set_mlp_mode(args.mlp_mode)
                extras["extract_mesh_sweep"] = extract_mesh_entry(kept[0], model, args.mlp_mode,
                                                                  f"sweep volume ({extras['sequence']['rows_end']} rows, 512^3)")
            del kept

        # ---- the widened rows: the global optimiser and whole-volume mesh extraction
        # on the bench's own volume.  LAST: the optimiser's Adam steps change the volume's features.
        if not args.no_widened and not tcnn:
            bnv.set_mlp_mode(args.mlp_mode)
            nm._drain_pipe()
            extras["extract_mesh"] = extract_mesh_entry(nm, model, args.mlp_mode,
                                                        f"bench volume ({nm.volume.num_rows()} rows, {args.grid}^3)")
            if args.input == "depth":
                extras["optimize"] = optimize_entry(nm, model, [frames[i] for i in pool[:32]], args.mlp_mode)

    m = args.mlp_mode
    peak = PEAK_TFLOPS[m]
    roof = dict(roofline_of(m, kern_run, kern_note), **({"power_limited_mfma_ceiling": power_ceiling} if power_ceiling else {}))
    out = dict(out_common)
    out.update({
        "value": main_run["fps"], "ms_per_step": 1e3 * main_run["elapsed"] / args.steps, "scaling": "weak",
        "value_note": f"the {args.steps} timed steps run right behind {args.preheat} untimed pre-heat frames + "
                      f"{args.warmup} warm-up steps (clock settled under the package power limit); `burst` is the same "
                      "steps from an idle GPU",
        "burst": entry(burst_run),
        "config": {"workload": workload, "grid": args.grid, "voxel_size": voxel, "preroll_frames": args.preroll,
                   "preheat_frames": args.preheat, "frames_per_step": 1, "mlp_mode": MODE_NAME[m],
                   "voxels_per_frame": main_run["n_vox"], "sdf_values_per_frame": 27.0 * main_run["n_vox"],
                   "decode_live_fraction": main_run["live"], "parallelism": "1 GPU",
                   "mlp_evals_last_timed_frame": main_run["rows"],
                   "persistent_lattice_tables": bool(getattr(getattr(nm, "_pipe", None), "persistent_tables", False)),
                   "mlp_evals_note": "evaluations of the last timed frame: with the persistent tables, entries of rows the "
                                     "frame did not update are carried over; roofline.mlp_evals_per_launch is a full "
                                     "recompute of a frame's entries (the kernel timed alone)"},
        "roofline": roof,
        "traffic_source": (roof["traffic_source"] or {}).get("file") if roof.get("traffic") else None,
        "kernels": {"pointnet_scatter": {"avg_ms": kern_run["enc_ms"], "tflops": kern_run["enc_tflops"],
                                         "frac_of_peak": kern_run["enc_tflops"] / peak}},
        "parity": main_run["parity"],
    })
    out.update(extras)
    # The figures a reader looks for first, in the objects the driver's record keeps whole (`config`, `roofline`) and once
    # more as the LAST key of the line (the driver also keeps the line's 2,000-character tail)
    summary = {"value_frames_per_s": out["value"], "roofline_frac": roof["frac"]}
    if "fp32_exact" in extras:
        summary["fp32_exact_frames_per_s"] = extras["fp32_exact"]["value"]
        summary["fp32_exact_roofline_frac_of_157.3TF"] = extras["fp32_exact"]["roofline"]["frac"]
        summary["fp32_exact_sdf_err"] = (extras["fp32_exact"].get("parity") or {}).get("sdf_max_abs_err_vs_oracle")
    if "tcnn_quick" in extras:
        summary["tcnn_frames_per_s_parity_unpinned"] = extras["tcnn_quick"]["value"]
    if "sustained" in extras:
        summary["sustained_1000_frames_per_s"] = extras["sustained"]["value"]
    if "optimize" in extras and isinstance(extras["optimize"], dict) and "value" in extras["optimize"]:
        summary["optimize_steps_per_s"] = extras["optimize"]["value"]
    out["config"]["also_measured"] = summary
    out["roofline"]["also_measured"] = {k: v for k, v in summary.items() if k.startswith("fp32_exact")}
    out["other_mlp_modes"] = [
        {"mlp_mode": MODE_NAME[a["mode"]], "dtype": DTYPE[a["mode"]], "value": a["fps"], "unit": "frames/s",
         "steps": a["steps"], "ms_per_step": 1e3 * a["elapsed"] / a["steps"], "decode_kernel_ms": a["dec_ms"],
         "decode_tflops": a["dec_tflops"], "decode_frac_of_peak": a["dec_tflops"] / PEAK_TFLOPS[a["mode"]],
         "pointnet_kernel_ms": a["enc_ms"], "pointnet_tflops": a["enc_tflops"], "parity": a["parity"]}
        for a in alts]
    if not args.no_cpu_baseline and not tcnn:
        out["cpu_baseline"] = cpu_baseline(depth_host[0], intr, synthetic.pose(0), args.grid)
        out["speedup_vs_cpu_baseline"] = main_run["fps"] / out["cpu_baseline"]["value"]
    out["summary"] = summary          # (last: inside the stored tail)
    return out


if __name__ == "__main__":
    main()
