#!/bin/bash
# One rank of a simulated world of 8 (real ghost rows), sustained, at --exchange-delay 0 and 30, with the stage timeline:
# the quick A/B of round 6's schedule work.  usage: tools/r06_shard_probe.sh <tag> [rank] [extra args]
TAG=$1; RANK=${2:-3}; shift; shift
O=gpurun_out/r06_probe_$TAG
mkdir -p $O
G=/tmp/bnv_ghosts_probe.pt
python tools/spatial_single_rank.py --world 8 --record $G > $O/record.txt 2>&1 || { tail -20 $O/record.txt; exit 1; }
python tools/spatial_single_rank.py --world 8 --rank 0 --ghosts $G --frames 300 --no-latency > /dev/null 2>&1   # warm the box
for D in 0 30; do
  python tools/spatial_single_rank.py --world 8 --rank $RANK --ghosts $G --frames 1500 --no-latency --exchange-delay $D --timeline 200 "$@" > $O/rank${RANK}_delay$D.txt 2>&1
  grep -h "pipelined wall clock\|MLP kernels\|durations:\|waits:\|across frames" $O/rank${RANK}_delay$D.txt | cut -c1-420
done
