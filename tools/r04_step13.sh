#!/bin/bash
set -u
O=gpurun_out/r04/s13
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
run() { N=$1; IF=$2; shift; shift
  env "$@" timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 1500 --in-flight $IF --no-latency 2>&1 | grep -v "$F" > $O/$N.txt
  echo "$N: $(grep 'pipelined' $O/$N.txt | cut -c1-150) | $(grep 'MLP kernels' $O/$N.txt)"
}
run s4_192_if3 3 BNV_PIPE_STREAMS=4
run s4_192_if5 5 BNV_PIPE_STREAMS=4
run s5_e64_t176_if4 4 BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=64 BNV_PIPE_TABLE_WGS=176
run s5_e64_t176_if5 5 BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=64 BNV_PIPE_TABLE_WGS=176
run s5_e72_t168_if5 5 BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=72 BNV_PIPE_TABLE_WGS=168
run s5_e80_t160_if5 5 BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=80 BNV_PIPE_TABLE_WGS=160
run s5_e192_tall_if5 5 BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=192 BNV_PIPE_TABLE_WGS=0
run s5_e96_t144_if5 5 BNV_PIPE_STREAMS=5 BNV_PIPE_ENCODER_WGS=96 BNV_PIPE_TABLE_WGS=144
