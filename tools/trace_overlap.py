"""Reads a rocprofv3 kernel trace (csv) of the pipelined loop: per frame wall time, sum of kernel durations, time some
kernel is running (union), time two kernels overlap; and one frame's timeline with the hardware queue of every launch.
    python tools/trace_overlap.py <kernel_trace.csv> [anchor-kernel-substring] [frames]"""
import csv
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_pointnet_scatter"
n_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 50
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"])
            for r in csv.DictReader(open(path)))
idx = [i for i, e in enumerate(ev) if anchor in e[2]]
lo, hi = ev[idx[-n_frames - 10]][0], ev[idx[-10]][0]
win = [e for e in ev if lo <= e[0] < hi]
tot = sum(e[1] - e[0] for e in win)
union, cs, ce = 0, None, None
for s, e, _, _ in win:
    if ce is None or s > ce:
        if ce is not None:
            union += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
print(f"{n_frames} frames: wall {(hi - lo) / n_frames / 1e3:.1f} us/frame, kernel durations {tot / n_frames / 1e3:.1f}, "
      f"some kernel running {union / n_frames / 1e3:.1f}, idle {(hi - lo - union) / n_frames / 1e3:.1f}")
i0 = idx[-20]
t0 = ev[i0][0]
for e in ev[i0 - 8:i0 + 22]:
    name = e[2].replace("bnv::", "").replace("void ", "").split("(")[0]
    print(f"{(e[0] - t0) / 1e3:9.1f} {(e[1] - t0) / 1e3:9.1f} {(e[1] - e[0]) / 1e3:7.1f}  q{e[3]}  {name[:40]}")
