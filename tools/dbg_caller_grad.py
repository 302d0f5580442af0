"""Debug: replay the caller trace and print how OUR gradients differ from the recorded ones, event by event."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bnv_fusion_amd as bnv
bnv.configure_runtime()
import test_gpu_caller_trace as T
z, meta, events = T._load()
rp = T.Replay(z, meta, tempfile.mkdtemp())
k = 0
for e in events:
    if e["depth"] != 0:
        continue
    op = e["op"]
    if op == "new": rp.new(e)
    elif op == "get": rp.get(e)
    elif op == "set": rp.set(e)
    elif op == "call":
        if e["method"] == "meshlize":
            rp.last_delta = rp.build(e["args"]["sdf_delta"])
        if e.get("caller_state"):
            for ref, d in e["caller_state"].items():
                ours = rp.refs[ref].detach().cpu().numpy()
                rec = z[d["data"]]
                print("caller_state", ref, "before applying: max |ours - recorded|", np.abs(ours - rec).max(), "rows changed",
                      int((np.abs(ours - rec).max(1) > 0).sum()))
        ret = rp.call(e)
        if e["method"] == "count_optim":
            w = rp.objs["volume"].weights.detach().cpu().numpy()
            print("after count_optim: weights sum", float(w.sum()), "rows >= 8:", int((w >= 8).sum()))
        if e["method"] == "decode_pts" and "data" in e["ret"]:
            ref = z[e["ret"]["data"]]
            o = ret.detach().cpu().numpy()
            print("decode_pts forward: max abs diff", np.abs(o - ref).max(), "masked ref", int((ref == np.float32(meta["voxel_size"])).sum()),
                  "ours", int((o == np.float32(meta["voxel_size"])).sum()))
    elif op == "grad":
        out, g_in = rp.pending
        leaf = rp.objs["volume"].features
        before = None if leaf.grad is None else leaf.grad.detach().clone()
        out.backward(g_in)
        got = (leaf.grad.detach() if before is None else leaf.grad.detach() - before).cpu().numpy()
        ref = z[e["value"]["data"]]
        d = np.abs(got - ref)
        off = d.max(1) > 1e-3 * np.abs(ref).max()
        print("grad event", k, "max|ref|", np.abs(ref).max(), "max diff", d.max(), "rows off", int(off.sum()),
              "rows nonzero ref", int((np.abs(ref).max(1) > 0).sum()), "got", int((np.abs(got).max(1) > 0).sum()),
              "only ref", int(((np.abs(ref).max(1) > 0) & (np.abs(got).max(1) == 0)).sum()),
              "only got", int(((np.abs(ref).max(1) == 0) & (np.abs(got).max(1) > 0)).sum()))
        if off.any():
            r = int(d.max(1).argmax())
            print("  worst row", r, "ref", ref[r], "\n  got", got[r], "\n  coord", rp.objs["volume"].active_coordinates[r].tolist(),
                  "weight", float(rp.objs["volume"].weights[r]))
        k += 1
