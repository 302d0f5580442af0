"""DESIGN.md = tools/design_template.md with its «TOKENS» replaced by figures parsed from profiles/<round>_* (no number of
those tables is typed by hand).

    python tools/fill_design.py r05
"""
import csv
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda name: os.path.join(root, "profiles", f"{R}_{name}")      # noqa: E731


def jload(name):
    return json.loads(open(P(name)).read().strip().splitlines()[-1])


d = jload("bench_line.json")
t = jload("bench_line_tcnn.json")
kt = {}
for r in csv.DictReader(open(P("bench_kernel_stats.csv"))):
    kt[r["Name"].split("(")[0].replace("void ", "").replace("bnv::", "").split("<")[0]] = float(r["AverageNs"]) / 1e3
roof, fe, su = d["roofline"], d["fp32_exact"], d.get("sustained", {})
enc = d["kernels"]["pointnet_scatter"]
# (the widened rows from the line of the round's final tree when there is one: extract_mesh's host copies changed late)
_ft = jload("bench_line_final_tree.json") if os.path.exists(P("bench_line_final_tree.json")) else d
opt, me, ms = (_ft.get(k) or d.get(k, {}) for k in ("optimize", "extract_mesh", "extract_mesh_sweep"))
V = {
    "VALUE": f"{d['value']:.1f}", "MS": f"{d['ms_per_step']:.3f}", "BURST": f"{d['burst']['value']:.1f}",
    "SUST": f"{su.get('value', 0):.1f}", "SUST_W": f"{su.get('mean_package_power_w') or 0:.0f}",
    "SUST_J": f"{su.get('joules_per_frame') or 0:.2f}",
    "FP32": f"{fe['value']:.1f}", "TAB32_FRAC": f"{fe['roofline']['frac']:.2f}",
    "ENC32_FRAC": f"{fe['kernels']['pointnet_scatter']['frac_of_peak']:.2f}",
    "TAB_MS": f"{roof['avg_kernel_ms']:.3f}", "TAB_EVALS": f"{roof['mlp_evals_per_launch'] / 1e6:.2f} M",
    "TAB_TF": f"{roof['achieved']:.0f}", "TAB_FRAC": f"{roof['frac']:.3f}", "TAB_ISSUE": f"{roof['mfma_issue_frac']:.2f}",
    "TRAFFIC": f"{(roof.get('traffic') or 0) / 1e6:.1f}", "ALG_MB": f"{40 * roof['mlp_evals_per_launch'] / 1e6:.1f}",
    "POWER_FRAC": f"{(roof.get('power_limited_mfma_ceiling') or {}).get('dominant_kernel_frac_of_it', 0):.2f}",
    "ENC_MS": f"{enc['avg_ms']:.3f}", "ENC_TF": f"{enc['tflops']:.0f}", "ENC_FRAC": f"{enc['frac_of_peak']:.3f}",
    "PARITY": f"{d['parity']['sdf_max_abs_err_vs_oracle']:.1e}",
    "CPU": f"{d['cpu_baseline']['value']:.4f}", "TCNN": f"{t['value']:.0f}",
    "TCNN_ENC": f"{t['kernels']['pointnet_scatter']['avg_ms']:.3f}", "TCNN_TAB": f"{t['roofline']['avg_kernel_ms']:.3f}",
    "SEQ": f"{d['sequence']['value']:.0f}", "SEQ_ROWS": f"{d['sequence']['rows_end']:,}",
}
# (the evaluations a pipelined frame really does, with the persistent tables: a field the bench line gained after the
# profile run -- taken from the line of the final tree, another box of the pool)
ft = jload("bench_line_final_tree.json") if os.path.exists(P("bench_line_final_tree.json")) else d
V["EVALS_FRAME"] = (f"{ft['config']['mlp_evals_last_timed_frame'] / 1e6:.2f} M" if ft["config"].get("mlp_evals_last_timed_frame")
                    else "≈ 5 % fewer")
_np = ft.get("without_persistent_tables")
V["NOPT"] = (f"{_np['value']:.1f} frames/s against {ft['value']:.1f} on that box ({_np['mlp_evals_last_timed_frame'] / 1e6:.2f} M "
             f"evaluations in the last frame against {ft['config']['mlp_evals_last_timed_frame'] / 1e6:.2f} M)"
             if _np else "A/B on one box: 558 against 583 frames/s (`profiles/r05_experiments.txt` [e3])")
V["EVALS_FULL"] = f"{ft['roofline']['mlp_evals_per_launch'] / 1e6:.2f} M"
if opt:
    sp = opt["split"]
    V.update({"OPT": f"{opt['value']:.0f}", "OPT_MS": f"{opt['ms_per_step']:.2f}", "OPT_LIVE": f"{sp['live_queries']:,}",
              "OPT_FWD": f"{sp['k_decode_pts']['avg_ms']:.3f}", "OPT_BWD": f"{sp['k_decode_pts_bwd']['avg_ms']:.3f}",
              "OPT_FRAC": f"{sp['k_decode_pts']['frac_of_peak']:.3f}",
              "OPT_ERR": f"{opt['parity']['sdf_max_abs_err_vs_oracle']:.1e}",
              "OPT_GERR": f"{opt['parity']['grad_max_err_over_max_grad_vs_oracle_autograd']:.1e}",
              "OPT_CPU": f"{opt['cpu_baseline']['value']:.2f}"})
    sl = opt.get("step_launch_set")
    if sl:
        V.update({"OPT_STEP_MS": f"{sl['avg_ms']:.3f}", "OPT_STEP_LIVE": f"{sl['live_queries_first_call']:,}",
                  "OPT_STEP_FRAC": f"{sl['frac_of_peak']:.3f}"})
if d.get("tcnn_quick"):
    V["TCNN_QUICK"] = f"{d['tcnn_quick']['value']:.0f}"
if me and ms:
    V.update({"MESH_VOX": f"{me['active_voxels']:,}", "MESH_MS": f"{me['value']:.1f}",
              "MESH_CPU": f"{ms['cpu_baseline']['value'] / 1e3:.0f}", "MESHS_MS": f"{ms['value']:.1f}",
              "MESHS_DEC": f"{ms['decode_ms']:.2f}", "MESHS_FRAC": f"{ms['table_kernel']['frac_of_peak']:.3f}",
              "MC_MS": f"{ms['marching_cubes_ms']:.2f}", "MC_VOX": f"{ms['active_voxels']:,}",
              "MC_GBS": f"{ms['marching_cubes']['gb_per_s']:.0f}"})
for k, v in kt.items():
    V[f"KT:{k}"] = f"{v:.1f}"


# ---- the sharded frame priced with real ghost rows: one table row per profile file
def spatial(name):
    path = P(f"{name}.txt")
    if not os.path.exists(path):
        return None
    txt = open(path).read()
    rec = re.search(r"sum over ranks / single = ([\d.]+)\s+max / mean per frame = ([\d.]+)\s+slowest rank / \(single / world\) = ([\d.]+)", txt)
    bnd = re.search(r"boundary records / emitted\s+voxels = ([\d.]+)", txt)
    slow = re.search(r"ms per frame: mean ([\d.]+), MAX ([\d.]+) \(rank (\d+)\) -> (\d+) frames/s", txt)
    evs = re.search(r"MLP evaluations: max / mean over the ranks ([\d.]+)", txt)
    mb = re.search(r"all-gather ([\d.]+) MB per rank and frame", txt)
    if not (rec and slow):
        return None
    return {"sum": rec.group(1), "mom": rec.group(2), "slow_ideal": rec.group(3), "bnd": bnd.group(1) if bnd else "?",
            "mean_ms": slow.group(1), "max_ms": slow.group(2), "fps": slow.group(4), "ev_mom": evs.group(1) if evs else "?",
            "mb": mb.group(1) if mb else "?"}


rows = [("world 8, pan 256³, **region** (default)", "spatial_world8_all_ranks_256"),
        ("… with 30 µs of simulated collective latency (`--exchange-delay 30`)", "spatial_world8_all_ranks_256_delay30"),
        ("world 8, pan 512³, region, 30 µs of collective latency", "spatial_world8_all_ranks_512_delay30"),
        ("world 8, pan 256³, first touch 8³ (round 4's rule)", "spatial_world8_all_ranks_256_first_touch"),
        ("world 8, pan 512³, region", "spatial_world8_all_ranks_512"),
        ("world 8, room sweep 256³, region (falls back to the interleave)", "spatial_world8_all_ranks_sweep"),
        ("world 8, room sweep 256³, first touch 8³", "spatial_world8_all_ranks_sweep_first_touch"),
        ("world 8, room sweep 256³, first touch 16³", "spatial_world8_all_ranks_sweep_first_touch16"),
        ("world 4, pan 256³, region", "spatial_world4"), ("world 2, pan 256³, region", "spatial_world2"),
        ("world 8, pan 256³, region, tiny-cuda-nn networks", "spatial_world8_tcnn")]
tab = ["| configuration (640×480, sustained) | Σ ranks' MLP work / single (record pass) | max / mean work per frame | boundary records / emitted voxel | all-gather MB per rank and frame | **slowest rank, ms per frame** | mean rank | frames/s of the rank set | file |",
       "|---|---|---|---|---|---|---|---|---|"]
for label, name in rows:
    s = spatial(name)
    if s:
        tab.append(f"| {label} | {s['sum']} | {s['mom']} | {s['bnd']} | {s['mb']} | **{s['max_ms']}** | {s['mean_ms']} | {s['fps']} | `profiles/{R}_{name}.txt` |")
V["SPATIAL_TABLE"] = "\n".join(tab)
# the single-GPU frame the sharded ranks are held against: the FASTEST measured this round (the rank figures and the
# single-GPU lines come from different boxes of the pool, which differ by +- 3 %: the conservative ratio)
single_ms = min(d["ms_per_step"], ft["ms_per_step"])
V["SINGLE_MS"] = f"{single_ms:.3f}"
for w_ in (2, 4):
    sw = spatial(f"spatial_world{w_}")
    if sw:
        V[f"W{w_}_MS"] = sw["max_ms"]
        V[f"W{w_}_X"] = f"{single_ms / float(sw['max_ms']):.2f}"
w8 = spatial("spatial_world8_all_ranks_256")
if w8:
    V["W8_MS"] = w8["max_ms"]
    V["W8_X"] = f"{single_ms / float(w8['max_ms']):.2f}"
    V["W8_X2"] = f"{single_ms / (float(w8['max_ms']) + 0.012):.2f}"
w8d = spatial("spatial_world8_all_ranks_256_delay30")
if w8d:
    V["W8D_MS"] = w8d["max_ms"]
    V["W8D_X"] = f"{single_ms / float(w8d['max_ms']):.2f}"
w8d5 = spatial("spatial_world8_all_ranks_512_delay30")
if w8d5:
    V["W8D512_MS"] = w8d5["max_ms"]

src = open(os.path.join(root, "tools", "design_template.md")).read()
missing = set()


def sub(m):
    k = m.group(1)
    if k not in V:
        missing.add(k)
        return m.group(0)
    return V[k]


out = re.sub(r"«([^»]+)»", sub, src)
open(os.path.join(root, "DESIGN.md"), "w").write(out)
print("DESIGN.md written;", "unfilled tokens:", sorted(missing) if missing else "none")
