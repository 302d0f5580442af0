#!/bin/bash
# (BNV_PIPE_CU_SHARED was a one-line experiment in pipeline.py -- both masks starting at CU 0 -- and is NOT in the tree: see profiles/r04_cu_mask_experiment.txt, finding 6)
mkdir -p gpurun_out/r04/s28
O=gpurun_out/r04/s28
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 300 --in-flight 3 --cu-split 0 --no-latency > /dev/null 2>&1
run() {
  N=$1; S=$2; shift; shift
  env "$@" timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 1000 --in-flight ${IF:-4} --cu-split $S --no-latency --timeline 100 2>&1 | grep -v "$F" > $O/$N.txt
  echo "$N: $(grep 'pipelined wall' $O/$N.txt | cut -c1-60) $(grep 'MLP kernels' $O/$N.txt | cut -c50-)"
  grep "durations\|across" $O/$N.txt | cut -c1-330
}
IF=3 run four 0 BNV_PIPE_STREAMS=4
for S in 224,224 192,192 224,192 224,128; do
  IF=4 run shared_${S} $S BNV_PIPE_CU_SHARED=1
  IF=3 run shared_${S}_if3 $S BNV_PIPE_CU_SHARED=1
done
