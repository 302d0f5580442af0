#!/bin/bash
set -u
O=gpurun_out/r04/s15
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=64 BNV_PIPE_TABLE_WGS=176
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o sp -- python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 300 --in-flight 5 --no-latency > $O/trace.log 2>&1
T=$(ls $O/tr/*/*kernel_trace.csv 2>/dev/null | head -1); [ -z "$T" ] && T=$(ls $O/tr/*kernel_trace.csv | head -1)
python3 tools/trace_overlap.py $T k_pointnet_scatter 60 2>&1 | head -60
rm -rf $O/tr
