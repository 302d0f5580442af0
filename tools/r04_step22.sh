#!/bin/bash
mkdir -p gpurun_out/r04/s22
O=gpurun_out/r04/s22
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 300 --in-flight 3 --cu-split 0 --no-latency > /dev/null 2>&1
for S in 192,64 160,64 0; do
  for IF in 5 4; do
  timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 1500 --in-flight $IF --cu-split $S --no-latency 2>&1 | grep -v "$F" > $O/split_${S}_if$IF.txt
  echo "split $S if $IF: $(grep 'pipelined wall' $O/split_${S}_if$IF.txt | cut -c1-60) $(grep 'MLP kernels' $O/split_${S}_if$IF.txt | cut -c50-)"
  grep "host time\|host enq" $O/split_${S}_if$IF.txt
  done
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 tools/spatial_single_rank.py --world 8 --rank 1 --frames 200 --in-flight 5 --cu-split 192,64 --no-latency > $O/trace.log 2>&1
ls $O/trace
