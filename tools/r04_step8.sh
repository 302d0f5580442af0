#!/bin/bash
set -u
O=gpurun_out/r04/s8
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python3 -m pytest tests/test_gpu_caller_trace.py -x -q 2>&1 | tail -15
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o sp -- python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 400 --no-latency --ownership first_touch > $O/trace.log 2>&1
T=$(ls $O/tr/*/*kernel_trace.csv 2>/dev/null | head -1); [ -z "$T" ] && T=$(ls $O/tr/*kernel_trace.csv | head -1)
python3 tools/trace_overlap.py $T k_pointnet_scatter 100 2>&1 | head -44
python3 - "$T" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_shard_assign" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("k_shard_assign durations (us): first 5", [round(x, 1) for x in d[:5]], "median", sorted(d)[len(d)//2], "max", max(d), "n", len(d))
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_shard_own" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("k_shard_own durations (us): median", sorted(d)[len(d)//2], "min", min(d), "max", max(d))
PY
rm -rf $O/tr
