#!/bin/bash
# One rank of a simulated world of 8 (real ghost rows), sustained, at --exchange-delay 0 and 30, under a list of settings
# (e.g. "--encoder-wgs 224"): the quick A/B of round 6's work on the sharded cycle.
#   tools/r06_shard_probe.sh <tag> <rank> "<args of config 1>" "<args of config 2>" ...
TAG=$1; RANK=$2; shift; shift
O=gpurun_out/r06_probe_$TAG
mkdir -p $O
G=/tmp/bnv_ghosts_probe.pt
python tools/spatial_single_rank.py --world 8 --record $G > $O/record.txt 2>&1 || { tail -20 $O/record.txt; exit 1; }
python tools/spatial_single_rank.py --world 8 --rank 0 --ghosts $G --frames 300 --no-latency > /dev/null 2>&1   # warm the box
K=0
for CFG in "$@"; do
  for D in 0 30; do
    F=$O/cfg${K}_delay$D.txt
    python tools/spatial_single_rank.py --world 8 --rank $RANK --ghosts $G --frames 1500 --no-latency --exchange-delay $D --timeline 200 $CFG > $F 2>&1
    echo "== [$CFG] delay $D: $(grep -h 'pipelined wall clock' $F | sed 's/->.*(10 seg/(10 seg/' | cut -c1-120)"
    grep -h "MLP kernels\|durations:\|across frames" $F | cut -c1-330
  done
  K=$((K+1))
done
