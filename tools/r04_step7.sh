#!/bin/bash
set -u
O=gpurun_out/r04/s7
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F="RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp\|amdgpu.ids"
for OWN in hash first_touch; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_$OWN -o sp -- python3 tools/spatial_single_rank.py --world 8 --rank 0 --frames 400 --no-latency --ownership $OWN > $O/trace_$OWN.log 2>&1
S=$(ls $O/tr_$OWN/*/*kernel_stats.csv 2>/dev/null | head -1); [ -z "$S" ] && S=$(ls $O/tr_$OWN/*kernel_stats.csv | head -1)
echo "== $OWN"; python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'].split('(')[0][-44:]:46s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
rm -rf $O/tr_$OWN
done
python3 - <<'PY'
# which grad event of the caller trace differs
import os, sys, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import bnv_fusion_amd as bnv
bnv.configure_runtime()
import test_gpu_caller_trace as T
z, meta, events = T._load()
rp = T.Replay(z, meta, tempfile.mkdtemp())
k = 0
for e in events:
    if e["depth"] != 0: continue
    op = e["op"]
    if op == "new": rp.new(e)
    elif op == "get": rp.get(e)
    elif op == "set": rp.set(e)
    elif op == "call":
        if e["method"] == "meshlize": rp.last_delta = rp.build(e["args"]["sdf_delta"])
        rp.call(e)
    elif op == "grad":
        out, g_in = rp.pending
        leaf = rp.objs["volume"].features
        before = None if leaf.grad is None else leaf.grad.detach().clone()
        out.backward(g_in)
        got = (leaf.grad.detach() if before is None else leaf.grad.detach() - before).cpu().numpy()
        ref = z[e["value"]["data"]]
        d = np.abs(got - ref)
        print("grad event", k, "max|ref|", np.abs(ref).max(), "max diff", d.max(), "rel", d.max() / np.abs(ref).max(),
              "grad before is None:", before is None, "rows off", int((d.max(1) > 1e-3 * np.abs(ref).max()).sum()))
        k += 1
    if k == 4: break
PY
