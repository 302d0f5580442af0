#!/bin/bash
set -u
O=gpurun_out/r04/s14
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
run() { N=$1; IF=$2; shift; shift
  env "$@" timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 1000 --in-flight $IF --no-latency 2>&1 | grep -v "$F" > $O/$N.txt
  echo "$N: $(grep 'pipelined' $O/$N.txt | cut -c1-100) | $(grep 'MLP kernels' $O/$N.txt | cut -c50-130) | $(grep 'pipeline streams' $O/$N.txt | cut -c80-250)"
}
run s5_e64_t176_q8 5 BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=64 BNV_PIPE_TABLE_WGS=176
run s5_e64_t176_q16 5 GPU_MAX_HW_QUEUES=16 BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=64 BNV_PIPE_TABLE_WGS=176
run s5_e64_t176_q24 5 GPU_MAX_HW_QUEUES=24 BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=64 BNV_PIPE_TABLE_WGS=176
run s4_q16 3 GPU_MAX_HW_QUEUES=16 BNV_PIPE_STREAMS=4
