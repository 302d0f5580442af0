#!/bin/bash
set -u
O=gpurun_out/r04/s12
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 900 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_determinism.py "tests/test_gpu_parity.py::test_hip_shards_equal_single_volume" tests/test_gpu_sequence.py -x -q 2>&1 | tail -6
run() { N=$1; shift
  env "$@" timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 1500 --in-flight 3 --no-latency 2>&1 | grep -v "$F" > $O/$N.txt
  echo "$N: $(grep 'pipelined' $O/$N.txt | cut -c1-150) | $(grep 'MLP kernels' $O/$N.txt)"
}
run s4_192 BNV_PIPE_STREAMS=4
run s5_192 BNV_PIPE_STREAMS=5
run s5_224 BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=224
run s5_all BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=0
run s5_160 BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=160
run s5_192_if4 BNV_PIPE_STREAMS=5 
