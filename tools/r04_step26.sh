#!/bin/bash
mkdir -p gpurun_out/r04/s26
O=gpurun_out/r04/s26
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 300 --in-flight 3 --no-latency > /dev/null 2>&1
for W in 144 160 176 192 208 224 240; do
  BNV_PIPE_ENCODER_WGS=$W timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 1500 --in-flight 3 --no-latency 2>&1 | grep -v "$F" > $O/enc_$W.txt
  echo "encoder on $W CUs: $(grep 'pipelined wall' $O/enc_$W.txt | cut -c1-60) $(grep 'MLP kernels' $O/enc_$W.txt | cut -c50-)"
done
for G in 512; do for W in 160 192 224; do
  BNV_PIPE_ENCODER_WGS=$W timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --grid $G --frames 1500 --in-flight 3 --no-latency 2>&1 | grep -v "$F" > $O/enc_${W}_g$G.txt
  echo "grid $G encoder on $W CUs: $(grep 'pipelined wall' $O/enc_${W}_g$G.txt | cut -c1-60) $(grep 'MLP kernels' $O/enc_${W}_g$G.txt | cut -c50-)"
done; done
