#!/bin/bash
set -u
O=gpurun_out/r04/s6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
DBG_MODE=1 timeout 300 python3 tools/dbg_caller_grad.py 2>&1 | grep -v "$F" | tail -12
DBG_MODE=0 timeout 300 python3 tools/dbg_caller_grad.py 2>&1 | grep -v "$F" | tail -8
for rep in 1 2; do for OWN in hash first_touch; do
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 1500 --in-flight 3 --no-latency --ownership $OWN 2>&1 | grep -v "$F" > $O/${OWN}_$rep.txt
echo "$OWN $rep: $(grep 'pipelined' $O/${OWN}_$rep.txt | cut -c1-140) | $(grep 'MLP kernels' $O/${OWN}_$rep.txt) | $(grep 'voxels owned' $O/${OWN}_$rep.txt | cut -c1-110)"
done; done
