#!/bin/bash
set -u
O=gpurun_out/r04/s11
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
one() { # tag, env..., -- args
  T=$1; shift
  env "$@" timeout 400 python3 bench.py --no-cpu-baseline --no-alt-mode --no-power-probe --sequence-frames 0 > $O/$T.json 2> $O/$T.err
  python3 - $O/$T.json $T <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "burst", round(d.get("burst", {}).get("value", 0), 1),
      "sustained", round(d.get("sustained", {}).get("value", 0), 1), "dec_ms", round(d["roofline"].get("kernel_ms", 0), 4) if "roofline" in d else None)
PY
}
one pipe_all BNV_PIPE_ENCODER_WGS=0
one pipe_248 BNV_PIPE_ENCODER_WGS=248
one pipe_240 BNV_PIPE_ENCODER_WGS=240
one pipe_224 BNV_PIPE_ENCODER_WGS=224
one stages BNV_NEURAL_MAP_PIPE=0
