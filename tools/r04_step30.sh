#!/bin/bash
# marking kernel: chunks of 1,024 lattice points per workgroup (product: 2) in the sharded frame
mkdir -p gpurun_out/r04/s30
O=gpurun_out/r04/s30
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 300 --in-flight 3 --no-latency > /dev/null 2>&1
for L in ${LIBS:-product tools/libbnv_mark_1024_1_0.so tools/libbnv_mark_1024_4_0.so product tools/libbnv_mark_1024_1_0.so}; do
  if [ $L = product ]; then unset BNV_FUSION_LIB; else export BNV_FUSION_LIB=$PWD/$L; fi
  timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 2000 --in-flight 3 --no-latency --timeline 200 2>&1 | grep -v "$F" > $O/$(basename $L).txt
  echo "$L: $(grep 'pipelined wall' $O/$(basename $L).txt | cut -c1-120)"
  grep "durations" $O/$(basename $L).txt | cut -c1-260
done
