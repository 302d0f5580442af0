#!/bin/bash
# round 4, step 2: per-call MLP modes (tests) + the encode stream a frame ahead of the host's bound wait
set -u
O=gpurun_out/r04/s2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 1200 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for A in 0 1; do for R in 0 1; do
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank $R --frames 2000 --in-flight 3 --ahead $A --no-latency 2>&1 | grep -v "$F" > $O/rank${R}_ahead$A.txt
done; done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o sp8 -- python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 400 --no-latency --ahead 1 > $O/trace_stdout.log 2>&1
T=$(ls $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1); [ -z "$T" ] && T=$(ls $O/trace/*kernel_trace.csv | head -1)
python3 tools/trace_overlap.py $T k_pointnet_scatter 100 > $O/overlap.txt 2>&1
rm -rf $O/trace
tail -5 $O/pytest.log; for f in $O/rank*; do echo $f; sed -n 2,6p $f; done; cat $O/overlap.txt | head -45
