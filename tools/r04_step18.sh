#!/bin/bash
set -u
O=gpurun_out/r04/s18
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
for Q in 16 24; do
GPU_MAX_HW_QUEUES=$Q timeout 600 python3 tools/spatial_single_rank.py --world 8 --all-ranks --frames 1000 --in-flight 3 2>&1 | grep -v "$F" > $O/all_q$Q.txt
echo "== queues $Q"; tail -14 $O/all_q$Q.txt | cut -c1-130
done
