#!/bin/bash
set -u
O=gpurun_out/r04/s16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -o b -- python3 bench.py --checkpoint tcnn --no-cpu-baseline --no-alt-mode --no-power-probe --sequence-frames 0 --preheat 200 > $O/tcnn_profiled.json 2> $O/tcnn.err
S=$(ls $O/tr/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$S" ] && S=$(ls $O/tr/*kernel_stats.csv | head -1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'].split('(')[0][-44:]:46s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
T=$(ls $O/tr/*/*kernel_trace.csv 2>/dev/null | head -1); [ -z "$T" ] && T=$(ls $O/tr/*kernel_trace.csv | head -1)
python3 tools/trace_overlap.py $T k_pointnet_scatter 40 2>&1 | head -36
rm -rf $O/tr
for S in 2 4; do
BNV_PIPE_STREAMS=$S timeout 300 python3 bench.py --checkpoint tcnn --no-cpu-baseline --no-alt-mode --no-power-probe --sequence-frames 0 > $O/tcnn_s$S.json 2>/dev/null
python3 - $O/tcnn_s$S.json $S <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tcnn streams", sys.argv[2], "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "burst", round(d.get("burst", {}).get("value", 0), 1))
PY
done
BNV_NEURAL_MAP_PIPE=0 timeout 300 python3 bench.py --checkpoint tcnn --no-cpu-baseline --no-alt-mode --no-power-probe --sequence-frames 0 > $O/tcnn_stages.json 2>/dev/null
python3 - $O/tcnn_stages.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tcnn per-stage path value", round(d["value"], 1), "ms", round(d["ms_per_step"], 4), "burst", round(d.get("burst", {}).get("value", 0), 1))
PY
