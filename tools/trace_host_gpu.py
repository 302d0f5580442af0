"""Host API calls against GPU kernel times from ONE `rocprofv3 --kernel-trace --hip-trace --output-format csv` run (round 6):
for a window of a few table-kernel cycles, every HIP call of the host thread (start, duration) and, for launches, the kernel
it started with its GPU start / end and the lag between the two -- which launches were queued ahead (large lag: the stream
was busy) and which the GPU was waiting for (lag ~5 us = launch latency).  The HIP-API trace slows the host (~6 us per call):
read the ORDER and the lags, not the cycle time.

  rocprofv3 --kernel-trace --hip-trace --output-format csv -d OUT -o t -- python3 tools/spatial_single_rank.py ...
  python3 tools/trace_host_gpu.py OUT/t_hip_api_trace.csv OUT/t_kernel_trace.csv [--anchor k_lattice_table_x] [--cycle 200] [--cycles 2]
"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("api")
ap.add_argument("kernels")
ap.add_argument("--anchor", default="k_lattice_table_x", help="kernel whose launches delimit the cycles")
ap.add_argument("--cycle", type=int, default=200, help="index of the anchor launch the window starts at")
ap.add_argument("--cycles", type=int, default=2)
args = ap.parse_args()

SKIP = {"__hipRegisterFunction", "hipGetDevice", "hipSetDevice", "hipGetLastError", "__hipPushCallConfiguration",
        "__hipPopCallConfiguration", "hipThreadExchangeStreamCaptureMode", "__hipRegisterFatBinary", "hipStreamIsCapturing",
        "hipStreamGetCaptureInfo", "hipGetDeviceCount"}
api = list(csv.DictReader(open(args.api)))
ker = list(csv.DictReader(open(args.kernels)))
by_corr = {r["Correlation_Id"]: r for r in ker}
ker.sort(key=lambda r: int(r["Start_Timestamp"]))
anchors = [r for r in ker if args.anchor in r["Kernel_Name"]]
t0 = int(anchors[args.cycle]["Start_Timestamp"])
t1 = int(anchors[args.cycle + args.cycles]["Start_Timestamp"])
print(f"us from the start of launch {args.cycle} of {args.anchor}; {args.cycles} cycles = {(t1 - t0) / 1e3:.1f} us")
print("    host  +dur  call                      -> kernel                      GPU start ..     end   (lag = GPU start - end of the call)")
out = []
for r in api:
    if r["Function"] in SKIP:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 - 150_000 or s > t1:
        continue
    k = by_corr.get(r["Correlation_Id"])
    extra = ""
    if k:
        ks, ke = int(k["Start_Timestamp"]), int(k["End_Timestamp"])
        name = k["Kernel_Name"].replace("void ", "").replace("bnv::", "")[:28]
        extra = f" -> {name:28s} {(ks - t0) / 1e3:8.1f} .. {(ke - t0) / 1e3:7.1f}   (lag {(ks - e) / 1e3:6.1f})"
    out.append((s, f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:5.1f}  {r['Function'][:24]:24s}{extra}"))
for _, line in sorted(out):
    print(line)
