#!/bin/bash
set -u
O=gpurun_out/r04/s17
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in 0 224 192 160 128 96; do
BNV_PIPE_ENCODER_WGS=$W timeout 300 python3 bench.py --checkpoint tcnn --no-cpu-baseline --no-alt-mode --no-power-probe --sequence-frames 0 > $O/tcnn_w$W.json 2>/dev/null
python3 - $O/tcnn_w$W.json $W <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tcnn encoder wgs", sys.argv[2], "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "burst", round(d.get("burst", {}).get("value", 0), 1), "enc ms", round(d["kernels"]["pointnet_scatter"]["avg_ms"], 4))
PY
done
