#!/bin/bash
# round 4, step 3: the four-stream frame pipeline (front / encode / main / blend), encoder on a share of the CUs
set -u
O=gpurun_out/r04/s3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 900 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_multiprocess.py tests/test_gpu_determinism.py tests/test_gpu_sequence.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
run() { # name, env...
  N=$1; shift
  env "$@" timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 1500 --in-flight 3 --no-latency 2>&1 | grep -v "$F" > $O/$N.txt
  echo "$N: $(grep 'pipelined' $O/$N.txt) | $(grep 'MLP kernels' $O/$N.txt)"
}
run s2 BNV_PIPE_STREAMS=2
run s4_all BNV_PIPE_STREAMS=4 BNV_PIPE_ENCODER_WGS=0
run s4_224 BNV_PIPE_STREAMS=4 BNV_PIPE_ENCODER_WGS=224
run s4_192 BNV_PIPE_STREAMS=4 BNV_PIPE_ENCODER_WGS=192
run s4_160 BNV_PIPE_STREAMS=4 BNV_PIPE_ENCODER_WGS=160
run s4_128 BNV_PIPE_STREAMS=4 BNV_PIPE_ENCODER_WGS=128
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o sp8 -- python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 400 --no-latency > $O/trace_stdout.log 2>&1
T=$(ls $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1); [ -z "$T" ] && T=$(ls $O/trace/*kernel_trace.csv | head -1)
python3 tools/trace_overlap.py $T k_pointnet_scatter 100 > $O/overlap.txt 2>&1
rm -rf $O/trace
cat $O/overlap.txt | head -45
