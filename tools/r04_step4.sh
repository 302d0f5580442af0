#!/bin/bash
# round 4, step 4: caller-trace replay + all ranks with the four-stream pipeline, 256^3 and 512^3
set -u
O=gpurun_out/r04/s4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 600 python3 -m pytest tests/test_gpu_caller_trace.py tests/test_gpu_modes.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -30 $O/pytest.log
timeout 900 python3 tools/spatial_single_rank.py --world 8 --all-ranks --frames 2000 --in-flight 3 2>&1 | grep -v "$F" > $O/all_ranks_256.txt
tail -14 $O/all_ranks_256.txt
timeout 900 python3 tools/spatial_single_rank.py --world 8 --all-ranks --frames 2000 --in-flight 3 --grid 512 2>&1 | grep -v "$F" > $O/all_ranks_512.txt
tail -14 $O/all_ranks_512.txt
