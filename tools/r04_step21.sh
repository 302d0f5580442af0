#!/bin/bash
# CU-masked five-stream schedule: symmetric splits only (multiples of 32 CUs = equal CUs per shader engine)
mkdir -p gpurun_out/r04/s21
O=gpurun_out/r04/s21
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 300 --in-flight 3 --cu-split 0 --no-latency > /dev/null 2>&1
for S in 0 160,64 192,64 128,64 160,96 128,96 192,32 160,32 0; do
  for IF in 3 4; do
  timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 1500 --in-flight $IF --cu-split $S --no-latency 2>&1 | grep -v "$F" > $O/split_${S}_if$IF.txt
  echo "split $S if $IF: $(grep 'pipelined wall' $O/split_${S}_if$IF.txt | cut -c1-60) $(grep 'MLP kernels' $O/split_${S}_if$IF.txt | cut -c50-)"
  done
done
