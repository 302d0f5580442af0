#!/bin/bash
# The multi-GPU part of tools/run_profiles.sh: the spatially sharded frame priced on one GPU, every rank, with the other
# ranks' real ghost rows (tools/spatial_single_rank.py).  bash tools/run_profiles_spatial.sh r05
set -u
R=${1:-r06}
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
# the spatially sharded frame priced on EVERY rank of a world of 8 WITH THE OTHER RANKS' REAL GHOST ROWS (record pass:
# all 8 shards in one process, real exchange; then every rank alone, timed, replaying the recorded blocks), sustained
# (2,000 frames per rank): the default ownership (contiguous regions) and round 4's fine interleave, the bench's pan at
# 256^3 and 512^3, the room sweep (a camera that turns and walks) under both rules; then worlds 2 and 4
SP="python3 tools/spatial_single_rank.py --all-ranks --in-flight 3"
$SP --world 8 --frames 2000 2>&1 | grep -v "$F" > $O/spatial_world8_all_ranks_256.txt
# ... and with 30 us of simulated collective latency behind the stand-in all-gather (VERDICT r05 item 1's condition)
$SP --world 8 --frames 2000 --exchange-delay 30 2>&1 | grep -v "$F" > $O/spatial_world8_all_ranks_256_delay30.txt
$SP --world 8 --frames 2000 --grid 512 --exchange-delay 30 2>&1 | grep -v "$F" > $O/spatial_world8_all_ranks_512_delay30.txt
$SP --world 8 --frames 2000 --ownership first_touch 2>&1 | grep -v "$F" > $O/spatial_world8_all_ranks_256_first_touch.txt
$SP --world 8 --frames 2000 --grid 512 2>&1 | grep -v "$F" > $O/spatial_world8_all_ranks_512.txt
$SP --world 8 --frames 1000 --scene sweep 2>&1 | grep -v "$F" > $O/spatial_world8_all_ranks_sweep.txt
$SP --world 8 --frames 1000 --scene sweep --ownership first_touch 2>&1 | grep -v "$F" > $O/spatial_world8_all_ranks_sweep_first_touch.txt
$SP --world 8 --frames 1000 --scene sweep --ownership first_touch --block-log2 4 2>&1 | grep -v "$F" > $O/spatial_world8_all_ranks_sweep_first_touch16.txt
$SP --world 2 --frames 1000 2>&1 | grep -v "$F" > $O/spatial_world2.txt
$SP --world 4 --frames 1000 2>&1 | grep -v "$F" > $O/spatial_world4.txt
$SP --world 8 --frames 2000 --checkpoint tcnn 2>&1 | grep -v "$F" > $O/spatial_world8_tcnn.txt
# GPU timestamps of every stage of 200 pipelined frames (bnv_frame_timeline), rank 1 with real ghosts: what the cycle consists of
python3 tools/spatial_single_rank.py --world 8 --record /tmp/gh_tl.pt > /dev/null 2>&1
python3 tools/spatial_single_rank.py --world 8 --ghosts /tmp/gh_tl.pt --rank 1 --frames 1000 --in-flight 3 --no-latency --timeline 200 2>&1 | grep -v "$F" > $O/spatial_world8_timeline.txt
python3 tools/spatial_single_rank.py --world 8 --ghosts /tmp/gh_tl.pt --rank 0 --frames 200 --in-flight 3 2>&1 | grep -v "$F" > $O/spatial_world8.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_sp8 -o sp8 -- python3 tools/spatial_single_rank.py --world 8 --ghosts /tmp/gh_tl.pt --rank 1 --frames 300 --no-latency > $O/trace_sp8_stdout.log 2>&1
