#!/bin/bash
# bench.py flag combinations the default line does not exercise: do they run and stay in parity?
Q="--no-cpu-baseline --no-alt-mode --no-power-probe --preheat 30 --steps 6 --warmup 2 --sustained-frames 0 --sequence-frames 0"
run() { echo "== $*"; timeout 600 python3 bench.py $Q "$@" 2>&1 | tail -1 | python3 -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('  value %.1f  parity %.2e  mask %s  dtype %s' % (d['value'], d['parity']['sdf_max_abs_err_vs_oracle'], d['parity']['mask_decisions_equal'], d['dtype'][:40]))
except Exception as e:
    print('  FAILED:', l[-400:])"; }
run --input points
run --sync-frames
run --no-stream-overlap
run --mlp-mode 0
run --mlp-mode 3
run --checkpoint tcnn --input points
run --checkpoint tcnn --sync-frames
run --grid 128 --input points
BNV_DIST_BACKEND=gloo run --gpus 2 --parallelism frame
BNV_NEURAL_MAP_PIPE=0 run
