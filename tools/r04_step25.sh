#!/bin/bash
# five streams WITHOUT masks, both MLP kernels on (almost) all CUs: they take turns by themselves (LDS), the chain of the
# next frame runs on the few CUs both leave free, beside whichever of them is running
mkdir -p gpurun_out/r04/s25
O=gpurun_out/r04/s25
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 300 --in-flight 3 --cu-split 0 --no-latency > /dev/null 2>&1
run() {  # name, env...
  N=$1; shift
  env "$@" timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 1500 --in-flight ${IF:-4} --cu-split 0 --no-latency 2>&1 | grep -v "$F" > $O/$N.txt
  echo "$N: $(grep 'pipelined wall' $O/$N.txt | cut -c1-60) $(grep 'MLP kernels' $O/$N.txt | cut -c50-)"
}
run four BNV_PIPE_STREAMS=4
for W in 240 224 208; do
  IF=4 run five_t${W}_e${W} BNV_PIPE_STREAMS=5 BNV_PIPE_TABLE_WGS=$W BNV_PIPE_ENCODER_WGS=$W
  IF=3 run five_t${W}_e${W}_if3 BNV_PIPE_STREAMS=5 BNV_PIPE_TABLE_WGS=$W BNV_PIPE_ENCODER_WGS=$W
done
IF=4 run five_t240_e192 BNV_PIPE_STREAMS=5 BNV_PIPE_TABLE_WGS=240 BNV_PIPE_ENCODER_WGS=192
IF=4 run five_t224_e192 BNV_PIPE_STREAMS=5 BNV_PIPE_TABLE_WGS=224 BNV_PIPE_ENCODER_WGS=192
IF=4 run five_t240_e240_lws2 BNV_PIPE_STREAMS=5 BNV_PIPE_TABLE_WGS=240 BNV_PIPE_ENCODER_WGS=240 BNV_PIPE_LATTICE_WS=2
run four_again BNV_PIPE_STREAMS=4
