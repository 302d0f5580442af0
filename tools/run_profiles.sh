#!/bin/bash
# Collects everything profiles/<round>_* is built from (run on the GPU box through gpurun; tools/make_profiles.py then
# builds the committed summaries from gpurun_out/).  Kernel trace and PMC passes are separate runs, as the pool requires.
set -u
R=${1:-r06}
O=gpurun_out/$R
mkdir -p $O
sha256sum bnv_fusion_amd/csrc/decode.hip > $O/decode_hip.sha256
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the default workload (50 steps, 5 warm-up), single stream, so that per-launch PMC figures match the default bench line
BENCH="python3 bench.py --no-cpu-baseline --no-alt-mode --no-stream-overlap --no-power-probe"
python3 bench.py > $O/bench_line.json 2> $O/bench_line.err
python3 bench.py --checkpoint tcnn --no-cpu-baseline > $O/bench_line_tcnn.json 2> $O/bench_line_tcnn.err
# the other BASELINE grids on one GPU (config 1: 128^3 / voxel 0.02; the 512^3 grid of config 3)
for G in 128 512; do python3 bench.py --grid $G --no-cpu-baseline --no-alt-mode > $O/bench_line_grid$G.json 2> $O/bench_line_grid$G.err; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $BENCH > $O/trace_stdout.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc1 -o p -- $BENCH > $O/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc2 -o p -- $BENCH > $O/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc3 -o p -- $BENCH > $O/pmc3.log 2>&1
# the tiny-cuda-nn (reference default) networks: kernel table + the scatter's write traffic
BENCHT="python3 bench.py --checkpoint tcnn --no-cpu-baseline --no-alt-mode --no-stream-overlap --preheat 100"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_tcnn -o bench -- $BENCHT > $O/trace_tcnn_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_tcnn_w -o p -- $BENCHT > $O/pmc_tcnn_w.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_tcnn_f -o p -- $BENCHT > $O/pmc_tcnn_f.log 2>&1
# the spatially sharded frame priced with real ghost rows (its own script: it can be re-run alone)
[ -n "${BNV_PROFILES_SKIP_SPATIAL:-}" ] || bash tools/run_profiles_spatial.sh $R
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
GPU_MAX_HW_QUEUES=4 python3 tools/queue_probe.py 2>&1 | grep "prio\|MAX" > $O/queue_probe.txt
python3 tools/mlp_launch_overhead.py 2>&1 | grep -v "$F" > $O/mlp_launch_overhead.txt
# the widened rows (global optimiser, whole-volume mesh extraction): their kernels in a trace of their own
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_widened -o w -- python3 tools/bench_optimize.py > $O/trace_widened_stdout.log 2>&1
# the optimiser loop alone: steps/s over 40 and 200 steps, the step's phases synchronised, and its kernel table
python3 tools/optimize_profile.py 2>&1 | grep -v "$F" > $O/optimize_profile.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_optimize -o opt -- python3 tools/optimize_profile.py > $O/trace_optimize_stdout.log 2>&1
python3 tools/fp_single_rank.py --replay 8 > $O/fp_replay8.txt 2>&1
# functional only: both multi-GPU decompositions as two gloo ranks sharing this GPU (no RCCL: it refuses two ranks per GPU)
BNV_DIST_BACKEND=gloo python3 bench.py --gpus 2 --no-cpu-baseline 2> $O/bench_line_2rank_gloo.err | tail -1 > $O/bench_line_2rank_gloo.json
# ... and as EIGHT gloo ranks sharing this GPU: real per-rank loads of a world of 8 (real exchange) in the line's per_rank
BNV_DIST_BACKEND=gloo python3 bench.py --gpus 8 --no-cpu-baseline --steps 20 --preheat 64 2> $O/bench_line_8rank_gloo.err | tail -1 > $O/bench_line_8rank_gloo.json
# the PMC summary of THIS run -> profiles/ of this copy, then the default bench line once more: its roofline.traffic comes from
# that summary (bench.py refuses one taken from another csrc/decode.hip); locally `python tools/make_profiles.py $R` rebuilds
# the same files from gpurun_out/
python3 tools/make_profiles.py $R > $O/make_profiles.log 2>&1
python3 bench.py > $O/bench_line_with_traffic.json 2> $O/bench_line_with_traffic.err
ls -la $O
