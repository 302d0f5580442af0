#!/bin/bash
# CU-masked five-stream schedule: correctness (pipeline / sharding tests), then a sweep of the split at world 8
mkdir -p gpurun_out/r04/s20
O=gpurun_out/r04/s20
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 900 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -x -q -m gpu -k "shard or pipe or async or Pipe" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for S in 0 168,64 0 176,64 160,72 168,72 160,80 152,88; do
  timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 1500 --in-flight 3 --cu-split $S --no-latency 2>&1 | grep -v "$F" > $O/split_$S.$RANDOM.txt; cp $(ls -t $O/split_$S.*.txt | head -1) $O/split_$S.txt
  echo "split $S: $(grep 'pipelined wall' $O/split_$S.txt)"
  grep "MLP kernels" $O/split_$S.txt
done
