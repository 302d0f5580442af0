#!/bin/bash
mkdir -p gpurun_out/r04/s27
O=gpurun_out/r04/s27
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 300 --in-flight 3 --cu-split 0 --no-latency > /dev/null 2>&1
run() {
  N=$1; shift
  env "$@" timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 1000 --in-flight ${IF:-4} --cu-split 0 --no-latency --timeline 100 2>&1 | grep -v "$F" > $O/$N.txt
  echo "$N: $(grep 'pipelined wall' $O/$N.txt | cut -c1-60) $(grep 'MLP kernels' $O/$N.txt | cut -c50-)"
  grep "durations\|waits:\|across" $O/$N.txt | cut -c1-400
}
IF=3 run four BNV_PIPE_STREAMS=4
for W in 224 192; do
  IF=4 run five_t${W}_e${W} BNV_PIPE_STREAMS=5 BNV_PIPE_TABLE_WGS=$W BNV_PIPE_ENCODER_WGS=$W
  IF=4 run five_t${W}_eall BNV_PIPE_STREAMS=5 BNV_PIPE_TABLE_WGS=$W BNV_PIPE_ENCODER_WGS=0
done
IF=4 run five_t224_e224_b BNV_PIPE_STREAMS=5 BNV_PIPE_TABLE_WGS=224 BNV_PIPE_ENCODER_WGS=224
