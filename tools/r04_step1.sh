#!/bin/bash
# round 4, step 1: baseline of the sharded frame on every rank of a simulated world of 8 + a two-stream kernel trace
set -u
O=gpurun_out/r04/s1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python3 tools/spatial_single_rank.py --world 8 --all-ranks --frames 2000 --in-flight 3 2>&1 | grep -v "$F" > $O/all_ranks_256.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o sp8 -- python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 400 --no-latency > $O/trace_stdout.log 2>&1
T=$(ls $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1); [ -z "$T" ] && T=$(ls $O/trace/*kernel_trace.csv | head -1)
python3 tools/trace_overlap.py $T k_pointnet_scatter 100 > $O/overlap.txt 2>&1
# keep only a window of the raw trace (the file is large)
python3 - "$T" $O/trace_window.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[-600:]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=["Kernel_Name", "Queue_Id", "Start_Timestamp", "End_Timestamp", "Workgroup_Size", "Grid_Size", "LDS_Block_Size", "VGPR_Count"], extrasaction="ignore")
w.writeheader()
for r in keep:
    r["Kernel_Name"] = r["Kernel_Name"].split("(")[0][-40:]
    w.writerow(r)
PY
rm -rf $O/trace
tail -3 $O/pytest.log; tail -14 $O/all_ranks_256.txt; cat $O/overlap.txt | head -50
