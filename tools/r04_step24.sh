#!/bin/bash
mkdir -p gpurun_out/r04/s24
O=gpurun_out/r04/s24
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 300 --in-flight 3 --cu-split 0 --no-latency > /dev/null 2>&1
for P in -1 0; do
for S in 192,64 160,64; do
  for IF in 2 3 4; do
  BNV_PIPE_MAIN_PRIORITY=$P timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 1500 --in-flight $IF --cu-split $S --no-latency 2>&1 | grep -v "$F" > $O/p${P}_split_${S}_if$IF.txt
  echo "prio $P split $S if $IF: $(grep 'pipelined wall' $O/p${P}_split_${S}_if$IF.txt | cut -c1-60) $(grep 'MLP kernels' $O/p${P}_split_${S}_if$IF.txt | cut -c50-)"
  done
done
done
