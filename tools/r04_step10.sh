#!/bin/bash
set -u
O=gpurun_out/r04/s10
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
for rep in 1 2; do for OWN in hash first_touch; do
timeout 300 python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 1500 --in-flight 3 --no-latency --ownership $OWN 2>&1 | grep -v "$F" > $O/${OWN}_$rep.txt
echo "$OWN $rep: $(grep 'pipelined' $O/${OWN}_$rep.txt | cut -c1-140) | $(grep 'MLP kernels' $O/${OWN}_$rep.txt) | $(grep 'voxels owned' $O/${OWN}_$rep.txt | cut -c1-110)"
done; done
timeout 600 python3 bench.py --no-cpu-baseline --steps 30 > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json | head -c 1500; echo
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/s10/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, "burst", d.get("burst", {}).get("value"), "sustained", d.get("sustained", {}).get("value"), "fp32", d.get("fp32_exact", {}).get("value"))
PY
timeout 600 python3 bench.py --no-cpu-baseline --checkpoint tcnn --steps 30 > $O/bench_tcnn.json 2> $O/bench_tcnn.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/s10/bench_tcnn.json").read().strip().splitlines()[-1])
print("tcnn", {k: d[k] for k in ("value", "ms_per_step")}, "burst", d.get("burst", {}).get("value"))
PY
