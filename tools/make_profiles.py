"""Builds profiles/<round>_* from gpurun_out/<round>/ after `tools/run_profiles.sh <round>` ran on the GPU box:

  bench_line.json, bench_line_tcnn.json   un-profiled bench lines
  trace/                                   rocprofv3 --kernel-trace --stats (single-stream bench)
  trace_stdout.log                         the bench line printed by that profiled run
  pmc1..3/                                 the three --pmc passes (SQ counters | FETCH_SIZE + GRBM_GUI_ACTIVE | WRITE_SIZE)
  spatial_world{8,2}.txt, fp_replay8.txt   multi-GPU modes priced on one GPU (tools/spatial_single_rank.py, fp_single_rank.py)

    python tools/make_profiles.py r02
"""
import collections, csv, glob, json, os, shutil, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = f"{root}/gpurun_out/{R}"
dst = f"{root}/profiles"


def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


def agg(d):
    f = glob.glob(f"{src}/{d}/**/*counter_collection.csv", recursive=True)[0]
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        a[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return a


stats = glob.glob(f"{src}/trace/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, f"{dst}/{R}_bench_kernel_stats.csv")
# (bench_line_with_traffic.json: `python bench.py` run again AFTER this script has written the PMC summary, so that its
# roofline.traffic is filled from it; preferred when present)
d = last_json(f"{src}/bench_line_with_traffic.json" if os.path.exists(f"{src}/bench_line_with_traffic.json") else f"{src}/bench_line.json")
dt = last_json(f"{src}/bench_line_tcnn.json")
dp = last_json(f"{src}/trace_stdout.log")
for name, obj in (("bench_line", d), ("bench_line_tcnn", dt), ("bench_line_profiled_run", dp)):
    open(f"{dst}/{R}_{name}.json", "w").write(json.dumps(obj) + "\n")
for t in ("spatial_world8", "spatial_world8_all_ranks_256", "spatial_world8_all_ranks_256_first_touch", "spatial_world8_all_ranks_512",
          "spatial_world8_all_ranks_256_delay30", "spatial_world8_all_ranks_512_delay30",
          "optimize_profile",
          "spatial_world8_all_ranks_sweep", "spatial_world8_all_ranks_sweep_first_touch", "spatial_world8_all_ranks_sweep_first_touch16",
          "spatial_world8_tcnn", "spatial_world8_timeline", "spatial_world2", "spatial_world4", "fp_replay8", "queue_probe",
          "mlp_launch_overhead"):
    if os.path.exists(f"{src}/{t}.txt"):
        import re
        keep = [l for l in open(f"{src}/{t}.txt").read().splitlines()
                if not re.search(r"RCCL version|HIP version|ROCm version|Hostname|Librccl|socket.cpp|amdgpu.ids", l)]
        open(f"{dst}/{R}_{t}.txt", "w").write("\n".join(keep) + "\n")
rows = list(csv.DictReader(open(f"{dst}/{R}_bench_kernel_stats.csv")))
# the dominant kernel's launches in time order: the rocprofv3 average over the SAME launches bench.py's HIP events cover
# (the last `steps` ones: the kernel-alone pass behind its pre-heat frames); the stats table averages over every launch
# of the run, and the kernel slows by ~20 % while the package heats up
trace_csv = glob.glob(f"{src}/trace/**/*kernel_trace.csv", recursive=True)
tail = {}
if trace_csv:
    tr = [r for r in csv.DictReader(open(trace_csv[0])) if "k_lattice_table_x" in r["Kernel_Name"]]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr]
    n_t = int(dp["steps"])
    tail = {"launches": len(du), "first_ms": sum(du[:n_t]) / max(len(du[:n_t]), 1), "last_ms": sum(du[-n_t:]) / max(len(du[-n_t:]), 1),
            "n": n_t}
pm = {}
with open(f"{dst}/{R}_pmc_summary.csv", "w") as f:
    f.write("kernel,counter,mean_per_launch,launches\n")
    for a in (agg("pmc1"), agg("pmc2"), agg("pmc3")):
        for k, v in a.items():
            if "bnv::" not in k:
                continue
            for c, x in v.items():
                f.write(f"\"{k}\",{c},{sum(x)/len(x):.6e},{len(x)}\n")
                pm[(k, c)] = sum(x) / len(x)
commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("BNV_COMMIT", "")   # (no .git on the GPU box: tools/run_profiles.sh passes it)
# the source the PMC passes saw (written on the GPU box by run_profiles.sh): bench.py drops the traffic figure when
# csrc/decode.hip no longer is that file
sha_path = f"{src}/decode_hip.sha256"
decode_sha = open(sha_path).read().split()[0] if os.path.exists(sha_path) else None
# (a rebuild of the summaries from the same gpurun_out/ keeps the commit the passes ran at; BNV_PROFILE_COMMIT names it
# when the passes were re-run at another commit)
if os.environ.get("BNV_PROFILE_COMMIT"):
    commit = os.environ["BNV_PROFILE_COMMIT"]
else:
  try:
    old = json.load(open(f"{dst}/{R}_pmc_meta.json"))
    if old.get("decode_hip_sha256") == decode_sha and old.get("mlp_evals_per_launch") == dp["roofline"]["mlp_evals_per_launch"]:
        commit = old.get("commit") or commit
  except (OSError, ValueError):
    pass
json.dump({"commit": commit, "decode_hip_sha256": decode_sha,
           "command": "python3 bench.py --no-cpu-baseline --no-alt-mode --no-stream-overlap --no-power-probe",
           "mlp_evals_per_launch": dp["roofline"]["mlp_evals_per_launch"], "dominant_kernel_trace": tail},
          open(f"{dst}/{R}_pmc_meta.json", "w"))


def g(sub, c):
    for (k, cc), v in pm.items():
        if sub in k and cc == c:
            return v
    return float("nan")


kt = {r["Name"].split("(")[0]: float(r["AverageNs"]) for r in rows}
calls = {r["Name"].split("(")[0]: int(r["Calls"]) for r in rows}
total_ns = sum(float(r["TotalDurationNs"]) for r in rows)


def kname(sub):
    c = [k for k in kt if k.replace("void ", "").replace("bnv::", "").split("(")[0].split("<")[0] == sub]
    return c[0] if c else None


lat = [k for k in kt if "k_lattice_table_x" in k][0]
enc = [k for k in kt if "k_pointnet_scatter_x" in k][0]
frame_kernels = ("k_front_mark", "k_rank", "k_finalize", "k_vol_integrate", "k_tsdf_integrate", "k_lattice_stamp",
                 "k_lattice_neighbors", "k_lattice_mark", "k_lattice_blend", "k_readback_words")
n_dec = calls[lat]
non_mlp_us = sum(kt[kname(s)] for s in frame_kernels if kname(s)) / 1e3
with open(f"{dst}/{R}_README.md", "w") as f:
    W = f.write
    W(f"# Round {int(R[1:])} profiles (one MI355X)\n\nGenerated by `tools/make_profiles.py {R}` from `tools/run_profiles.sh {R}` (commit {commit}).\n\n")
    fe = d.get("fp32_exact", {})
    su = d.get("sustained", {})
    gr = d.get("growth", {})
    sq = d.get("sequence", {})
    W("* `%s_bench_line.json` -- `python bench.py` (defaults: 50 steps, 5 warm-up behind 1,000 pre-heat frames; fp32 checkpoint, split_f16 MLP mode): `value` **%.1f frames/s** (%.3f ms/frame; a sustained rate), the same steps from an idle GPU (`burst`) %.1f; `sustained` pass over %d frames %.1f frames/s at %.0f MHz / %.0f W (%.2f J per frame); exact-fp32 MFMA mode %.1f frames/s (decode kernel %.3f of the fp32 MFMA peak); growing from the reference's 100,000-row capacity %.1f frames/s over the first %d frames; moving-camera `sequence` (%d frames, 512^3, growth inside the loop) %.1f frames/s to %d rows, oracle checks %s; parity vs the oracle on %d voxels: SDF max-abs-err %.1e (bar 1e-4), mask decisions equal: %s; CPU oracle baseline %.4f frames/s (%d threads of %d cores).\n" % (
        R, d["value"], d["ms_per_step"], d.get("burst", {}).get("value", 0), su.get("frames", 0), su.get("value", 0), su.get("mean_sclk_mhz") or 0, su.get("mean_package_power_w") or 0, su.get("joules_per_frame") or 0,
        fe.get("value", 0), fe.get("roofline", {}).get("frac", 0), gr.get("value", 0), gr.get("frames", 0),
        sq.get("frames", 0), sq.get("value", 0), sq.get("rows_end", 0), ", ".join("%.1e" % c["sdf_max_abs_err"] for c in sq.get("parity_checks", sq.get("oracle_checks", []))),
        d["parity"]["voxels_checked"], d["parity"]["sdf_max_abs_err_vs_oracle"], d["parity"]["mask_decisions_equal"],
        d.get("cpu_baseline", {}).get("value", 0), d.get("cpu_baseline", {}).get("cores", 0), d.get("cpu_baseline", {}).get("host_cores", 0)))
    W(f"* `{R}_bench_line_tcnn.json` -- the same with `--checkpoint tcnn` (the reference's default tiny-cuda-nn networks): {dt['value']:.1f} frames/s sustained ({dt['ms_per_step']:.3f} ms/frame), {dt.get('burst', {}).get('value', 0):.1f} burst; encoder {dt['kernels']['pointnet_scatter']['avg_ms']:.3f} ms, lattice-table kernel {dt['roofline']['avg_kernel_ms']:.3f} ms.\n")
    tst = glob.glob(f"{src}/trace_tcnn/**/*kernel_stats.csv", recursive=True)
    if tst:
        shutil.copy(tst[0], f"{dst}/{R}_bench_kernel_stats_tcnn.csv")
        tw = glob.glob(f"{src}/pmc_tcnn_w/**/*counter_collection.csv", recursive=True)
        tf = glob.glob(f"{src}/pmc_tcnn_f/**/*counter_collection.csv", recursive=True)
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for fl in tw + tf:
            for r_ in csv.DictReader(open(fl)):
                acc[r_["Kernel_Name"].split("(")[0]][r_["Counter_Name"]].append(float(r_["Counter_Value"]))
        with open(f"{dst}/{R}_pmc_summary_tcnn.csv", "w") as ft:
            ft.write("kernel,counter,mean_per_launch,launches\n")
            for k, v in acc.items():
                if "bnv::" in k:
                    for c, x in v.items():
                        ft.write(f"\"{k}\",{c},{sum(x)/len(x):.6e},{len(x)}\n")
        enc_t = [k for k in acc if "k_pointnet_scatter_tb" in k]
        if enc_t and "WRITE_SIZE" in acc[enc_t[0]]:
            wkb = sum(acc[enc_t[0]]["WRITE_SIZE"]) / len(acc[enc_t[0]]["WRITE_SIZE"])
            W(f"* `{R}_bench_kernel_stats_tcnn.csv`, `{R}_pmc_summary_tcnn.csv` -- kernel table and WRITE_SIZE / FETCH_SIZE passes of `bench.py --checkpoint tcnn --no-stream-overlap`: the block encoder `k_pointnet_scatter_tb` writes **{wkb * 1024 / 1e6:.0f} MB per launch** (one LDS table per workgroup and 16x16-pixel patch; one table per wave and 8x4 block: 101 MB; round 2's per-tile scatter: 366 MB, every device-scope atomic tallied at 64 B).\n")
    sp8 = glob.glob(f"{src}/trace_sp8/**/*kernel_stats.csv", recursive=True)
    if sp8:
        shutil.copy(sp8[0], f"{dst}/{R}_spatial_world8_kernel_stats.csv")
        W(f"* `{R}_spatial_world8_kernel_stats.csv` -- rocprofv3 kernel table of `tools/spatial_single_rank.py --world 8` (rank 0 of a simulated world of 8; the profiler serialises the two streams).\n")
    for G in (128, 512):
        if os.path.exists(f"{src}/bench_line_grid{G}.json"):
            dg = last_json(f"{src}/bench_line_grid{G}.json")
            open(f"{dst}/{R}_bench_line_grid{G}.json", "w").write(json.dumps(dg) + "\n")
            W(f"* `{R}_bench_line_grid{G}.json` -- `python bench.py --grid {G}` (voxel {dg['config']['voxel_size']}): {dg['value']:.1f} frames/s, {dg['config']['voxels_per_frame']:.0f} voxels decoded per frame, parity on {dg['parity']['voxels_checked']} voxels {dg['parity']['sdf_max_abs_err_vs_oracle']:.1e}, mask decisions equal: {dg['parity']['mask_decisions_equal']}.\n")
    W(f"* `{R}_bench_kernel_stats.csv` -- `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-alt-mode --no-stream-overlap --no-power-probe` (the default workload: 30 fuse-only pre-roll frames + 55 fuse+decode frames; one stream, so every kernel's duration is its own).  `{R}_bench_line_profiled_run.json` is the line that very run printed: its HIP-event average for the dominant kernel (the last {tail.get('n', 0)} launches: the kernel-alone pass behind its pre-heat frames) is {dp['roofline']['avg_kernel_ms']:.3f} ms; rocprofv3's kernel trace over the same {tail.get('n', 0)} launches: **{tail.get('last_ms', 0):.3f} ms**; the table's {kt[lat]/1e6:.3f} ms averages all {tail.get('launches', 0)} launches of the run, whose first {tail.get('n', 0)} (idle GPU) take {tail.get('first_ms', 0):.3f} ms -- the package heats up and the clock settles ~20 % lower.\n")
    W(f"* `{R}_pmc_summary.csv` (+ `{R}_pmc_meta.json`: commit and MLP evaluations per launch of the profiled run) -- three separate `rocprofv3 --pmc ... --kernel-trace` passes of that command (SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA | FETCH_SIZE GRBM_GUI_ACTIVE | WRITE_SIZE), mean per launch.\n")
    def tail_of(name, n=3):
        pth = f"{dst}/{R}_{name}.txt"
        return " | ".join(l.strip() for l in open(pth).read().splitlines()[-n:] if l.strip()) if os.path.exists(pth) else "(not collected)"
    W(f"* `{R}_spatial_world8_all_ranks_256.txt`, `{R}_spatial_world8_all_ranks_512.txt` -- `tools/spatial_single_rank.py --world 8 --all-ranks --frames 2000`: the spatially sharded frame through the product path (C frame pipeline, four streams, region ownership) priced on EVERY rank of a world of 8 WITH THE OTHER RANKS' REAL GHOST ROWS (a record pass runs all 8 shards in one process with the real exchange and keeps every frame's blocks; every rank then runs alone, timed, replaying them), 2,000 frames per rank (sustained), each rank's voxels / pairs / MLP evaluations next to the record pass's; the rank set runs at the pace of its slowest rank.  256^3: {tail_of('spatial_world8_all_ranks_256', 4)}.  512^3: {tail_of('spatial_world8_all_ranks_512', 4)}.\n")
    W(f"* `{R}_spatial_world8_all_ranks_256_first_touch.txt` -- the same with round 4's fine-interleave first touch: {tail_of('spatial_world8_all_ranks_256_first_touch', 4)}.\n")
    W(f"* `{R}_spatial_world8_all_ranks_sweep.txt`, `..._sweep_first_touch.txt`, `..._sweep_first_touch16.txt` -- the room sweep (`sequence.py`: a camera that turns and walks), 256^3, 1,000 frames per rank, under the region rule, the fine interleave with 8^3 blocks and with 16^3 blocks.  region: {tail_of('spatial_world8_all_ranks_sweep', 4)}.  first touch 8^3: {tail_of('spatial_world8_all_ranks_sweep_first_touch', 4)}.  first touch 16^3: {tail_of('spatial_world8_all_ranks_sweep_first_touch16', 4)}.\n")
    W(f"* `{R}_spatial_world2.txt`, `{R}_spatial_world4.txt` (worlds 2 and 4, every rank), `{R}_spatial_world8.txt` (rank 0, 200 frames from an idle GPU + the latency of one frame at a time), `{R}_spatial_world8_tcnn.txt` (tiny-cuda-nn networks); `{R}_fp_replay8.txt` -- `tools/fp_single_rank.py --replay 8`: the frame-parallel mode's per-batch work of one rank of 8.\n")
    if os.path.exists(f"{src}/bench_line_8rank_gloo.json") and os.path.getsize(f"{src}/bench_line_8rank_gloo.json") > 10:
        d8 = last_json(f"{src}/bench_line_8rank_gloo.json")
        open(f"{dst}/{R}_bench_line_8rank_gloo_functional.json", "w").write(json.dumps(d8) + "\n")
        pr = d8.get("spatial_sharding", {}).get("per_rank", [])
        W(f"* `{R}_bench_line_8rank_gloo_functional.json` -- `BNV_DIST_BACKEND=gloo python bench.py --gpus 8 --steps 20 --preheat 64`: EIGHT ranks sharing this one GPU over gloo (functional: its rates mean nothing), the real exchange among 8 shards: `spatial_sharding.per_rank[].mlp_evals_per_frame` = {[int(x['mlp_evals_per_frame']) for x in pr]}; parity of rank 0's outputs against the oracle {d8.get('spatial_sharding', {}).get('parity', {}).get('sdf_max_abs_err_vs_oracle')}.\n")
    W(f"* `{R}_spatial_world8_all_ranks_256_delay30.txt`, `{R}_spatial_world8_all_ranks_512_delay30.txt` -- the same with `--exchange-delay 30` (a 30 us spin kernel behind the stand-in all-gather: the latency of a real 8-rank collective): {tail_of('spatial_world8_all_ranks_256_delay30', 3)} || 512^3: {tail_of('spatial_world8_all_ranks_512_delay30', 3)}.\n")
    ost = glob.glob(f"{src}/trace_optimize/**/*kernel_stats.csv", recursive=True)
    if ost:
        shutil.copy(ost[0], f"{dst}/{R}_optimize_kernel_stats.csv")
        W(f"* `{R}_optimize_profile.txt`, `{R}_optimize_kernel_stats.csv` -- `tools/optimize_profile.py` (the global optimiser's loop alone, 5,000 rays per step in 5 splits): steps/s over 40 and 200 steps, the phases of a step with a device synchronisation behind each, and the rocprofv3 kernel table of the same script (`k_optim_step` = forward + L1 loss + backward of all five splits in one launch): {tail_of('optimize_profile', 7)}.\n")
    wst = glob.glob(f"{src}/trace_widened/**/*kernel_stats.csv", recursive=True)
    if wst:
        shutil.copy(wst[0], f"{dst}/{R}_widened_kernel_stats.csv")
        W(f"* `{R}_widened_kernel_stats.csv` -- rocprofv3 kernel table of `tools/bench_optimize.py`: the global optimiser's kernels (`k_decode_pts`, `k_decode_pts_bwd`, `k_ray_*`, `k_vol_count_optim*`) and whole-volume mesh extraction (`k_mc_count_indexed`, `k_mc_emit_indexed`, the lattice kernels on every active voxel): the rows behind bench.py's `optimize` / `extract_mesh` entries.\n")
    if os.path.exists(f"{src}/bench_line_2rank_gloo.json") and os.path.getsize(f"{src}/bench_line_2rank_gloo.json") > 10:
        d2 = last_json(f"{src}/bench_line_2rank_gloo.json")
        open(f"{dst}/{R}_bench_line_2rank_gloo_functional.json", "w").write(json.dumps(d2) + "\n")
        W(f"* `{R}_bench_line_2rank_gloo_functional.json` -- `BNV_DIST_BACKEND=gloo python bench.py --gpus 2`: FUNCTIONAL evidence only (two ranks share one GPU, the collectives go through the host): the line a multi-GPU run prints -- `value` = the spatially sharded mode ({d2['scaling']}), `spatial_sharding` and `frame_parallel` entries with their own parity checks (spatial: SDF max-abs-err {d2['spatial_sharding']['parity']['sdf_max_abs_err_vs_oracle']:.1e}, mask decisions equal: {d2['spatial_sharding']['parity']['mask_decisions_equal']}), `distributed` = what the backend saw.  Its rates say nothing about RCCL over xGMI.\n")
    W(f"* `{R}_queue_probe.txt` -- `GPU_MAX_HW_QUEUES=4 tools/queue_probe.py`: which pool streams overlap the default stream (every fourth shares its hardware queue and runs strictly behind it); `{R}_mlp_launch_overhead.txt` -- `tools/mlp_launch_overhead.py`: duration of the two MLP kernels against the work per launch, down to an empty launch.\n\n")
    n_small = sum(1 for s in frame_kernels if kname(s))
    W(f"A frame is **{n_small + 2} kernel launches** ({n_small} small kernels + the two MLP kernels; the upsert also stamps the decode's origins and the frame's counters come back through a read-back launch, not through copies); the kernels that are not an MLP take **{non_mlp_us:.0f} us** per frame together.\n\n")
    W("| kernel | calls | avg ms | % of GPU time |\n|---|---|---|---|\n")
    for r in [r for r in rows if "k_probe_mfma" not in r["Name"]][:14]:
        W(f"| `{r['Name'].split('(')[0][:64]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['Percentage']):.2f} |\n")
    W("\n| kernel | MFMA pipe busy | waves parked (WAIT_ANY) | issue-stalled (WAIT_INST_ANY) | issuing | FETCH_SIZE KB (x2 for wide reads, MI355X_MICROARCH.md) | WRITE_SIZE KB | sustained clock |\n|---|---|---|---|---|---|---|---|\n")
    for name, sub, key in (("k_lattice_table_x<3> (lattice table, split_f16)", "k_lattice_table_x<3>", lat), ("k_pointnet_scatter_x<3>", "k_pointnet_scatter_x", enc)):
        gui = g(sub, "GRBM_GUI_ACTIVE") / 8
        wc = g(sub, "SQ_WAVE_CYCLES")
        W(f"| `{name}` | {100*g(sub,'SQ_VALU_MFMA_BUSY_CYCLES')/(1024*gui):.0f} % | {100*g(sub,'SQ_WAIT_ANY')/wc:.0f} % | {100*g(sub,'SQ_WAIT_INST_ANY')/wc:.0f} % | {100*g(sub,'SQ_ACTIVE_INST_ANY')/wc:.0f} % | {g(sub,'FETCH_SIZE'):.0f} | {g(sub,'WRITE_SIZE'):.0f} | {gui/kt[key]:.2f} GHz |\n")
    W("\nGather / scatter kernels (HBM-bound side of the path): achieved HBM rate = (2 x FETCH_SIZE + WRITE_SIZE) per launch / average duration from the kernel-trace table, against the 8 TB/s peak.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide reads on gfx950; for narrow gathers and for WRITE_SIZE (which also counts every device-scope atomic as a 64-B write to the memory side) the absolute is uncalibrated -- read the column as an upper estimate.  These kernels move a few MB each; at 10-40 us they are bound by launch + dependent-access latency, not by bandwidth.\n\n")
    W("| kernel | avg us | FETCH KB | WRITE KB | HBM MB / launch | achieved GB/s | of 8 TB/s |\n|---|---|---|---|---|---|---|\n")
    for sub in frame_kernels:
        key = kname(sub)
        if not key or (key, "FETCH_SIZE") not in pm:
            continue
        fe_, wr = pm[(key, "FETCH_SIZE")], pm[(key, "WRITE_SIZE")]
        mb = (2 * fe_ + wr) * 1024 / 1e6
        gbs = mb * 1e6 / kt[key]
        W(f"| `{sub}` | {kt[key]/1e3:.1f} | {fe_:.0f} | {wr:.0f} | {mb:.1f} | {gbs:.0f} | {gbs/8000:.3f} |\n")
    r = d["roofline"]
    W("\nDominant kernel `k_lattice_table_x<3>`: %.3f ms (HIP events, kernel-alone pass of bench.py) for %.3g MLP evaluations x 402,432 FLOP = %.3f TFLOP -> %.0f TFLOP/s algorithmic = %.3f of the 2.5 PFLOP/s f16 MFMA peak; the split issues 3 MFMA products per algorithmic product, i.e. %.2f of peak MFMA issue.  HBM traffic per launch (2 x FETCH + WRITE) = %.1f MB against %.1f MB algorithmic (40 B x evaluations): MFMA-bound, weights stream from L2.\n" % (
        r["avg_kernel_ms"], r["mlp_evals_per_launch"], r["flop_per_launch"] / 1e12, r["achieved"], r["frac"], r["mfma_issue_frac"],
        (2 * g("k_lattice_table_x<3>", "FETCH_SIZE") + g("k_lattice_table_x<3>", "WRITE_SIZE")) * 1024 / 1e6, 40 * dp["roofline"]["mlp_evals_per_launch"] / 1e6))
    pl = r.get("power_limited_mfma_ceiling")
    if pl:
        W("\n**The package power limit, not the clock, is the ceiling.**  An MFMA-only stream (`bnv_probe_mfma_rate`, every CU, two waves per SIMD, ~8 ms, run by bench.py right behind the timed frames) sustains %.0f TFLOP/s of `v_mfma_f32_16x16x32_f16` and %.0f TFLOP/s of `v_mfma_f32_32x32x16_f16` with random f16 operands on this box (%.0f with zero operands) against the 2,500 the clock would allow; the dominant kernel issues %.0f TFLOP/s of MFMA products = **%.2f of that ceiling**.\n" % (
            pl["tflops_16x16x32_random_f16_operands"], pl["tflops_32x32x16_random_f16_operands"], pl["tflops_16x16x32_zero_operands"],
            pl["dominant_kernel_issued_tflops"], pl["dominant_kernel_frac_of_it"]))
    extra = []
    for t, what in (("power_probe", "`tools/power_probe.py`: both MLP kernels with their real weights and with zeroed ones (identical instruction streams: the difference is power) + the MFMA-only rates"),
                    ("probe_shapes", "`tools/probe_shapes.hip`: MFMA-only streams of the two f16 MFMA shapes, 1 / 2 / 4 rotating operand sets"),
                    ("probe_fillers", "`tools/probe_fillers.hip`: instructions of each kind that fit in the shadow of an MFMA; operand-data dependence of the MFMA rate")):
        if os.path.exists(f"{src}/{t}.txt"):
            shutil.copy(f"{src}/{t}.txt", f"{dst}/{R}_{t}.txt")
            extra.append(f"* `{R}_{t}.txt` -- {what}.")
    if extra:
        W("\n" + "\n".join(extra) + "\n")
    # records written by hand or by their own tools (kept as they are; listed when present)
    notes = []
    for t, what in (("spatial_world8_timeline", "`tools/spatial_single_rank.py --timeline 200`: GPU timestamps of every stage of the sharded frame (`bnv_frame_timeline`): what the cycle consists of"),
                    ("cu_mask_probe", "`tools/cu_mask_probe.py`: CU-masked HIP streams partition the GPU; disjoint masks run side by side"),
                    ("cu_mask_experiment", "the sharded frame on CU-masked streams (five streams, table MLP from a feature snapshot): every split measured, all slower than four streams; why"),
                    ("fifth_stream_experiment", "the same without masks (earlier in the round)"),
                    ("experiments", "the smaller experiments DESIGN.md cites (round 3's tree on every rank, its kernel trace window, the single-GPU frame through the pipeline)"),
                    ("soak", "`tools/soak_pipeline.py`: 1,500-3,000 frames through the four-stream pipeline with three frames in flight against the same frames one at a time (one GPU fp32 / tcnn; rank 1 of a simulated world of 8): every frame's outputs and the final volume bit-identical")):
        if os.path.exists(f"{dst}/{R}_{t}.txt"):
            notes.append(f"* `{R}_{t}.txt` -- {what}.")
    if os.path.exists(f"{dst}/{R}_bench_line_final_tree.json"):
        ft = last_json(f"{dst}/{R}_bench_line_final_tree.json")
        npt = ft.get("without_persistent_tables")
        notes.append(f"* `{R}_bench_line_final_tree.json` -- `python bench.py` of the round's final tree (another box than "
                     f"`{R}_bench_line.json`; same kernels): {ft['value']:.1f} frames/s"
                     + (f"; the A/B inside that run with the persistent lattice tables off (`without_persistent_tables`): "
                        f"{npt['value']:.1f} frames/s, {npt['mlp_evals_last_timed_frame'] / 1e6:.2f} M evaluations per frame "
                        f"against {ft['config']['mlp_evals_last_timed_frame'] / 1e6:.2f} M" if npt else "") + ".")
    if notes:
        W("\n" + "\n".join(notes) + "\n")
print(open(f"{dst}/{R}_README.md").read()[:3000])
