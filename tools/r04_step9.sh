#!/bin/bash
set -u
O=gpurun_out/r04/s9
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 900 python3 -m pytest tests/test_gpu_caller_trace.py "tests/test_gpu_parity.py::test_hip_shards_equal_single_volume" tests/test_gpu_pipeline.py -x -q 2>&1 | tail -12
for rep in 1 2; do for OWN in hash first_touch; do
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 1500 --in-flight 3 --no-latency --ownership $OWN 2>&1 | grep -v "$F" > $O/${OWN}_$rep.txt
echo "$OWN $rep: $(grep 'pipelined' $O/${OWN}_$rep.txt | cut -c1-140) | $(grep 'MLP kernels' $O/${OWN}_$rep.txt) | $(grep 'voxels owned' $O/${OWN}_$rep.txt | cut -c1-110)"
done; done
timeout 1200 python3 -m pytest tests/test_gpu_multiprocess.py -x -q 2>&1 | tail -5
