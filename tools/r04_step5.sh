#!/bin/bash
# round 4, step 5: first-touch ownership + caller trace
set -u
O=gpurun_out/r04/s5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 900 python3 -m pytest tests/test_gpu_caller_trace.py "tests/test_gpu_parity.py::test_hip_shards_equal_single_volume" -x -q > $O/pytest1.log 2>&1; echo "pytest rc $?" >> $O/pytest1.log
tail -40 $O/pytest1.log
timeout 1200 python3 -m pytest tests/test_gpu_multiprocess.py tests/test_gpu_pipeline.py -x -q > $O/pytest2.log 2>&1; echo "pytest rc $?" >> $O/pytest2.log
tail -15 $O/pytest2.log
timeout 900 python3 tools/spatial_single_rank.py --world 8 --all-ranks --frames 2000 --in-flight 3 2>&1 | grep -v "$F" > $O/all_ranks_256.txt
tail -16 $O/all_ranks_256.txt
