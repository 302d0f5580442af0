"""Prints a window of a `rocprofv3 --kernel-trace --output-format csv` trace as a timeline: every kernel between the
start of the N-th last table kernel and the end of the (N - span)-th last, with its hardware queue, stream, grid and
duration -- what runs beside what in the sharded frame's cycle (profiles/r05_experiments.txt [e8] was read off this).

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 tools/spatial_single_rank.py --world 8 \
        --ghosts gh.pt --rank 3 --frames 300 --no-latency
    python tools/trace_window.py out/<host>/<pid>_kernel_trace.csv [--back 8] [--span 2] [--anchor k_lattice_table_x]
"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--back", type=int, default=8, help="start at the back-th last launch of the anchor kernel")
ap.add_argument("--span", type=int, default=2, help="anchor launches covered")
ap.add_argument("--anchor", default="k_lattice_table_x")
args = ap.parse_args()
rows = list(csv.DictReader(open(args.trace)))


def name(r):
    return r["Kernel_Name"].replace("void ", "").replace("bnv::", "").split("(")[0][:44]


anchors = [i for i, r in enumerate(rows) if args.anchor in r["Kernel_Name"]]
i0, i1 = anchors[-args.back], anchors[-args.back + args.span]
t0 = int(rows[i0]["Start_Timestamp"])
sel = [r for r in rows if t0 - 30000 <= int(r["Start_Timestamp"]) <= int(rows[i1]["End_Timestamp"]) + 20000]
sel.sort(key=lambda r: int(r["Start_Timestamp"]))
print("  start us    end us      us  queue stream   workgroups x threads   kernel")
for r in sel:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    wgs = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    print(f"{s:10.1f} {e:9.1f} {e - s:7.1f}  q{r['Queue_Id']:>2}   s{r['Stream_Id']:>3}   {wgs:>6} x {r['Workgroup_Size_X']:>4}   {name(r)}")
