"""What an event record / wait between two kernels of ONE stream costs on this stack (round 6).  Chains of 100-us spin kernels
(bnv_probe_spin: one wave) on a stream, with nothing / an event record / a wait for an event that has long fired / a record
that another stream waits for between them; run under `rocprofv3 --kernel-trace --output-format csv` and read the gaps
between consecutive kernels of the stream from the trace (tools/probe_event_gap.py --read <kernel_trace.csv>).
Answers whether the 18 us behind the table kernel of a sharded frame is the price of the packets
between the kernels."""
import argparse
import csv
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--read", default=None)
args = ap.parse_args()

CASES = ["nothing", "event record", "wait (fired event of another stream)", "record + another stream waits and runs a kernel",
         "record + wait(fired) + wait(fired)", "big kernel (256 x 512 threads) then nothing", "big kernel then event record",
         "big kernel, record + other stream waits and runs",
         "wait for an event another stream records WHILE the kernel in front runs",
         "record + wait for an event another stream records while the kernel in front runs",
         "two records"]
REP = 6

if args.read:
    rows = [r for r in csv.DictReader(open(args.read)) if "k_probe_spin" in r["Kernel_Name"] or "k_probe_fill" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the main stream's kernels come in groups of REP + 1 per case, separated by > 5 ms
    main_q = rows[0]["Queue_Id"]
    main = [r for r in rows if r["Queue_Id"] == main_q]
    groups, cur = [], [main[0]]
    for a, b in zip(main, main[1:]):
        if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 2_000_000:
            groups.append(cur)
            cur = []
        cur.append(b)
    groups.append(cur)
    side = [r for r in rows if r["Queue_Id"] != main_q]
    for name, g in zip(CASES, groups[1:]):          # (group 0: warm-up)
        gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(g, g[1:])]
        print(f"{name:80s} gaps between consecutive kernels of the stream (us): " + " ".join(f"{x:5.1f}" for x in gaps))
        if "other stream waits" in name:             # the hop: end of the main stream's kernel -> start of the waiter's
            hops = []
            for a in g[:-1]:
                nxt = [int(r["Start_Timestamp"]) for r in side if int(r["Start_Timestamp"]) >= int(a["End_Timestamp"])]
                if nxt:
                    hops.append((min(nxt) - int(a["End_Timestamp"])) / 1e3)
            print(f"{'':80s} main kernel ends -> the waiting stream's kernel starts (us):   " + " ".join(f"{x:5.1f}" for x in hops))
    sys.exit(0)

import time
import torch
from bnv_fusion_amd import _lib
from bnv_fusion_amd.streams import concurrent_stream

lib = _lib.load()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
M = torch.cuda.current_stream()
S = concurrent_stream(dev, M)
ptr = lambda s: C.c_void_p(s.cuda_stream)      # noqa: E731
CY = 210_000                             # ~100 us


def spin(s, blocks=1, cycles=CY):
    _lib.check(lib.bnv_probe_spin(blocks, cycles, ptr(s)), "spin")


fired = torch.cuda.Event()
with torch.cuda.stream(S):
    spin(S)
    fired.record(S)
torch.cuda.synchronize()
for _ in range(4):
    spin(M)
torch.cuda.synchronize()
time.sleep(0.01)
evs = [torch.cuda.Event() for _ in range(64)]
for case in range(len(CASES)):
    big = case >= 5
    for k in range(REP + 1):
        spin(M, 256 * 8 if big else 1)   # (one wave per block: 2,048 blocks fill every CU's wave slots of one SIMD row)
        if k == REP:
            break
        e = evs[(case * REP + k) % 64]
        if case in (1, 6):
            e.record(M)
        elif case == 2:
            M.wait_event(fired)
        elif case in (3, 7):
            e.record(M)
            S.wait_event(e)
            spin(S, 1, 20_000)
        elif case == 4:
            e.record(M)
            M.wait_event(fired)
            M.wait_event(fired)
        elif case in (8, 9):
            # (enqueued BEFORE the main stream's kernel above has finished: the host is ahead, as in the pipeline; the
            # other stream's kernel is short, so its event fires while that kernel still runs)
            if case == 9:
                e.record(M)
            spin(S, 1, 20_000)
            e2 = evs[(case * REP + k + 32) % 64]
            e2.record(S)
            M.wait_event(e2)
        elif case == 10:
            e.record(M)
            evs[(case * REP + k + 32) % 64].record(M)
    torch.cuda.synchronize()
    time.sleep(0.01)
print("done")
