#!/bin/bash
# encoder CU share at world 2 and 4 (world 8: flat between 144 and 192, tools/r04_step26.sh)
mkdir -p gpurun_out/r04/s31
O=gpurun_out/r04/s31
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 300 --in-flight 3 --no-latency > /dev/null 2>&1
for W in 2 4; do for E in 160 192 224 240 0; do
  BNV_PIPE_ENCODER_WGS=$E timeout 300 python3 tools/spatial_single_rank.py --world $W --rank 1 --frames 1000 --in-flight 3 --no-latency 2>&1 | grep -v "$F" > $O/w${W}_e$E.txt
  echo "world $W encoder on $E CUs: $(grep 'pipelined wall' $O/w${W}_e$E.txt | cut -c1-60) $(grep 'MLP kernels' $O/w${W}_e$E.txt | cut -c50-)"
done; done
