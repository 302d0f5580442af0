#!/bin/bash
set -u
O=gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
python3 tools/spatial_single_rank.py --world 8 --all-ranks --frames 2000 --in-flight 3 2>&1 | grep -v "$F" > $O/spatial_world8_all_ranks_256.txt
tail -15 $O/spatial_world8_all_ranks_256.txt | cut -c1-150
python3 tools/spatial_single_rank.py --world 8 --all-ranks --frames 2000 --in-flight 3 --grid 512 2>&1 | grep -v "$F" > $O/spatial_world8_all_ranks_512.txt
tail -15 $O/spatial_world8_all_ranks_512.txt | cut -c1-150
python3 tools/spatial_single_rank.py --world 8 --all-ranks --frames 2000 --in-flight 3 --ownership hash 2>&1 | grep -v "$F" > $O/spatial_world8_all_ranks_256_hash.txt
tail -15 $O/spatial_world8_all_ranks_256_hash.txt | cut -c1-150
python3 tools/spatial_single_rank.py --world 2 --all-ranks --frames 1000 --in-flight 3 2>&1 | grep -v "$F" > $O/spatial_world2.txt
tail -8 $O/spatial_world2.txt | cut -c1-150
