"""DESIGN.md = tools/design_template.md with its @@tokens@@ replaced by figures read from the committed profiles
(profiles/<round>_*): every number in the design document then has a file behind it.
    python tools/fill_design.py [r04]"""
import csv, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = lambda n: os.path.join(root, "profiles", f"{R}_{n}")      # noqa: E731
d = json.loads(open(P("bench_line.json")).read())
t = json.loads(open(P("bench_line_tcnn.json")).read())
ks = {r["Name"].split("(")[0].replace("void ", "").replace("bnv::", ""): r for r in csv.DictReader(open(P("bench_kernel_stats.csv")))}
kt = {r["Name"].split("(")[0].replace("void ", "").replace("bnv::", ""): r for r in csv.DictReader(open(P("bench_kernel_stats_tcnn.csv")))}


def us(table, sub):
    c = [k for k in table if k.split("<")[0] == sub or k == sub]
    c = c or [k for k in table if k.startswith(sub)]
    k = max(c, key=lambda k: int(table[k]["Calls"]))
    return float(table[k]["AverageNs"]) / 1e3


def ranks(name):
    rows, out = [], {}
    for l in open(P(name)).read().splitlines():
        m = re.match(r"\s+(\d+)\s+([\d.]+)\s+\(([\d.]+), ([\d.]+)\)\s+(\d+)\s+(\d+)\s+(\d+)", l)
        if m:
            rows.append([float(x) for x in m.groups()])
        m = re.match(r"\s+(voxels owned|pairs encoded|MLP evaluations): max / mean over the ranks ([\d.]+)", l)
        if m:
            out[m.group(1)] = float(m.group(2))
    ms = [r[1] for r in rows]
    out.update(max=max(ms), mean=sum(ms) / len(ms), fps=1e3 / max(ms), n=len(rows))
    return out


su, fe, ro = d["sustained"], d["fp32_exact"], d["roofline"]
ft256, ft512, hs = ranks("spatial_world8_all_ranks_256.txt"), ranks("spatial_world8_all_ranks_512.txt"), ranks("spatial_world8_all_ranks_256_hash.txt")
single_ms = d["ms_per_step"]
tc_sp = re.search(r"pipelined wall clock\s+([\d.]+) ms", open(P("spatial_world8_tcnn.txt")).read()).group(1)
fp = open(P("fp_replay8.txt")).read()
m = re.search(r"([\d.]+) ms per batch", fp)
ratio = single_ms / ft256["max"]
tok = {
    "k_front_mark": f"{us(ks, 'k_front_mark'):.1f} µs", "k_rank": f"{us(ks, 'k_rank'):.1f} µs",
    "k_pointnet_scatter_x": f"**{us(ks, 'k_pointnet_scatter_x') / 1e3:.3f} ms** = {d['kernels']['pointnet_scatter']['tflops']:.0f} TFLOP/s algorithmic = {d['kernels']['pointnet_scatter']['frac_of_peak']:.3f} of 2.5 PF (bench line, sustained); exact fp32: {fe['kernels']['pointnet_scatter']['frac_of_peak']:.2f} of 157.3 TF",
    "k_finalize": f"{us(ks, 'k_finalize'):.1f} µs", "k_vol_integrate": f"{us(ks, 'k_vol_integrate<true>'):.1f} µs",
    "k_lattice_neighbors": f"{us(ks, 'k_lattice_neighbors'):.1f} µs", "k_lattice_mark": f"{us(ks, 'k_lattice_mark'):.1f} µs",
    "k_lattice_table_x": f"**{us(ks, 'k_lattice_table_x') / 1e3:.3f} ms** mean of all launches of the profiled run; sustained, timed alone by bench.py: {ro['avg_kernel_ms']:.3f} ms for {ro['mlp_evals_per_launch'] / 1e6:.2f} M evaluations = {ro['achieved']:.0f} TFLOP/s = **{ro['frac']:.3f}** of 2.5 PF ({ro['mfma_issue_frac']:.2f} of peak MFMA issue: 3 products per algorithmic product); exact fp32: **{fe['roofline']['frac']:.2f}** of 157.3 TF",
    "k_lattice_blend": f"{us(ks, 'k_lattice_blend'):.1f} µs", "k_tsdf_integrate": f"{us(ks, 'k_tsdf_integrate'):.1f} µs",
    "k_pointnet_scatter_tb": f"{us(kt, 'k_pointnet_scatter_tb'):.0f} µs", "k_lattice_table_t": f"{us(kt, 'k_lattice_table_t'):.0f} µs",
    "value": f"**{d['value']:.1f}** frames/s ({d['ms_per_step']:.3f} ms/frame, {ro['mlp_evals_per_launch'] / 1e6:.2f} M MLP evaluations per frame)",
    "burst": f"{d['burst']['value']:.1f}", "sustained": f"{su['value']:.1f} frames/s at {su.get('mean_sclk_mhz') or 0:.0f} MHz, {su.get('mean_package_power_w') or 0:.0f} W, **{su.get('joules_per_frame') or 0:.2f} J per frame**",
    "fp32_exact": f"{fe['value']:.1f} frames/s; decode kernel {fe['roofline']['achieved']:.1f} TFLOP/s = **{fe['roofline']['frac']:.3f}** of the 157.3 TFLOP/s fp32 MFMA peak, encoder {fe['kernels']['pointnet_scatter']['frac_of_peak']:.2f}",
    "roofline": f"{ro['flop_per_launch'] / 1e12:.3f} TFLOP algorithmic per launch ÷ {ro['avg_kernel_ms']:.3f} ms = {ro['achieved']:.0f} TFLOP/s = **{ro['frac']:.3f}** of the 2.5 PF f16 peak; the box's MFMA-only ceiling (`power_limited_mfma_ceiling`): " + (f"{ro['power_limited_mfma_ceiling']['tflops_16x16x32_random_f16_operands']:.0f} TFLOP/s, the kernel issues {ro['power_limited_mfma_ceiling']['dominant_kernel_frac_of_it']:.2f} of it" if ro.get("power_limited_mfma_ceiling") else "n/a"),
    "traffic": (f"{ro['traffic'] / 1e6:.1f} MB/launch against {40 * ro['mlp_evals_per_launch'] / 1e6:.1f} MB algorithmic (40 B × evaluations)" if ro.get("traffic") else "not in this line (the PMC passes did not exist yet when it ran); `profiles/%s_README.md`: 2 × FETCH + WRITE per launch of the profiled run" % R),
    "parity": f"{d['parity']['sdf_max_abs_err_vs_oracle']:.1e}", "mode3": ", ".join(f"{o['value']:.0f} frames/s, SDF error {o['parity']['sdf_max_abs_err_vs_oracle']:.1e}" for o in d.get("other_mlp_modes", [])),
    "cpu": f"{d['cpu_baseline']['value']:.4f} frames/s ({d['cpu_baseline']['cores']} threads of the host's {d['cpu_baseline']['host_cores']} cores; ≈ {1 / d['cpu_baseline']['value']:.0f} s per frame, of which the 216-evaluations-per-voxel decode is {d['cpu_baseline']['decode_s_scaled']:.0f} s)",
    "tcnn": f"**{t['value']:.0f}** frames/s sustained ({t['ms_per_step']:.3f} ms/frame), {t['burst']['value']:.0f} burst; encoder {t['kernels']['pointnet_scatter']['avg_ms']:.3f} ms, table kernel {t['roofline']['avg_kernel_ms']:.3f} ms",
    "frame_sum": f"{ro['avg_kernel_ms']:.3f} + {d['kernels']['pointnet_scatter']['avg_ms']:.3f} ms of MLP kernels in a {d['ms_per_step']:.3f} ms frame; the nine other launches take {sum(us(ks, k) for k in ('k_front_mark', 'k_rank', 'k_finalize', 'k_vol_integrate<true>', 'k_tsdf_integrate', 'k_lattice_neighbors', 'k_lattice_mark', 'k_lattice_blend', 'k_readback_words')):.0f} µs alone (`profiles/{R}_bench_kernel_stats.csv`)",
    "balance": f"{ft256['voxels owned']:.3f} / {ft256['pairs encoded']:.3f} / {ft256['MLP evaluations']:.3f} at 256³ and {ft512['voxels owned']:.3f} / {ft512['pairs encoded']:.3f} / {ft512['MLP evaluations']:.3f} at 512³ (block hash: {hs['voxels owned']:.3f} / {hs['pairs encoded']:.3f} / {hs['MLP evaluations']:.3f})",
    "hash_max": f"{hs['max']:.3f}", "hash_mean": f"{hs['mean']:.3f}", "hash_fps": f"{hs['fps']:,.0f}",
    "ft256_max": f"{ft256['max']:.3f}", "ft256_mean": f"{ft256['mean']:.3f}", "ft256_fps": f"{ft256['fps']:,.0f}",
    "ft512_max": f"{ft512['max']:.3f}", "ft512_mean": f"{ft512['mean']:.3f}", "ft512_fps": f"{ft512['fps']:,.0f}",
    "tcnn_sp": f"{tc_sp}", "single_ms": f"{single_ms:.3f}", "ratio": f"{ratio:.1f}",
    "verdict": ("met by the one-GPU pricing without the collective's real latency (%.1f× with a 30 µs all-gather on top)" % (single_ms / (ft256['max'] + 0.03)))
    if ratio >= 6.0 else ("NOT met (%.1f×; %.1f× with a 30 µs all-gather on top)" % (ratio, single_ms / (ft256['max'] + 0.03))),
    "fp_replay": (m.group(1) + " ms") if m else "see the file",
}
tok["exposed"] = (f"{1e3 * (d['ms_per_step'] - ro['avg_kernel_ms'] - d['kernels']['pointnet_scatter']['avg_ms']):.0f} µs with the fp32 checkpoint "
                  f"({d['ms_per_step']:.3f} ms frame − {ro['avg_kernel_ms']:.3f} − {d['kernels']['pointnet_scatter']['avg_ms']:.3f}), "
                  f"{1e3 * (t['ms_per_step'] - t['roofline']['avg_kernel_ms'] - t['kernels']['pointnet_scatter']['avg_ms']):.0f} µs with the tiny-cuda-nn networks "
                  f"({t['ms_per_step']:.3f} − {t['roofline']['avg_kernel_ms']:.3f} − {t['kernels']['pointnet_scatter']['avg_ms']:.3f})")
w2, w4 = ranks("spatial_world2.txt"), ranks("spatial_world4.txt")
tok["curve"] = ("| ranks | slowest rank, ms per frame | frames/s of the rank set | × one GPU | file |\n|---|---|---|---|---|\n"
                f"| 1 | {single_ms:.3f} | {1e3 / single_ms:,.0f} | 1 | `profiles/{R}_bench_line.json` (`value`) |\n"
                f"| 2 | {w2['max']:.3f} | {w2['fps']:,.0f} | {single_ms / w2['max']:.2f} | `profiles/{R}_spatial_world2.txt` |\n"
                f"| 4 | {w4['max']:.3f} | {w4['fps']:,.0f} | {single_ms / w4['max']:.2f} | `profiles/{R}_spatial_world4.txt` |\n"
                f"| 8 | {ft256['max']:.3f} | {ft256['fps']:,.0f} | {single_ms / ft256['max']:.2f} | `profiles/{R}_spatial_world8_all_ranks_256.txt` |")
tl = open(P("spatial_world8_timeline.txt")).read()
N = r"([\d.]+)"


def g(pat):
    return float(re.search(pat.replace("#", N), tl).group(1))


v = {"up": g(r"finalize [\d.]+, upsert #,"), "ex": g(r"exchange \(host-enqueued\) #"), "ins": g(r"[\d.]+, install #"), "mt": g(r"marking \+ table #"),
     "gap": g(r"table\(t\) done -> upsert\(t\+1\) starts #"), "cyc": g(r"cycle \(table done to table done\) #"),
     "wall": g(r"pipelined wall clock\s+# ms") * 1e3, "enc": g(r"encoder #, finalize"), "slack": g(r"finalize done -> upsert starts #"),
     "front": g(r"front end #")}
tok["timeline"] = (
    "upsert {up:.0f} µs → exchange {ex:.0f} (this tool's stand-in: one RCCL call on a one-rank group + three small launches) → "
    "install {ins:.0f} → marking + table {mt:.0f} (table kernel ≈ 150: 6 rounds of 24 µs + 16 µs where 5.25 rounds of work exist, "
    "§3.7) → {gap:.0f} µs until the next frame's upsert starts = the cycle ({cyc:.0f} µs with the timeline's marker events, {wall:.0f} "
    "without).  The encoder ({enc:.0f} µs on 192 CUs, sharing the GPU with other frames' kernels) runs two frames ahead: finalize "
    "is done {slack:.0f} µs before its upsert starts, the front end ({front:.0f} µs) before that").format(**v)
src = open(os.path.join(root, "tools", "design_template.md")).read()
missing = sorted(set(re.findall(r"@@(\w+)@@", src)) - set(tok))
assert not missing, missing
out = re.sub(r"@@(\w+)@@", lambda mm: tok[mm.group(1)], src)
open(os.path.join(root, "DESIGN.md"), "w").write(out)
print("DESIGN.md written;", len(tok), "figures from profiles/%s_*" % R)
