"""Soak of the four-stream frame pipeline: N frames pipelined (3 in flight) against the same frames one at a time, from an
empty volume each -- every frame's outputs (coordinates, SDF lattice) and the final volume must be bit-identical.  A race
between the streams (a slot reused too early, a workspace hazard) shows up as a differing checksum.
    python3 tools/soak_pipeline.py --frames 2000 [--world 8 --rank 1]
"""
import argparse, hashlib, os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bnv_fusion_amd as bnv
bnv.configure_runtime()
from bnv_fusion_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=2000)
ap.add_argument("--world", type=int, default=1)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--checkpoint", default="fp32")
args = ap.parse_args()
dims, voxel = synthetic.GRID_DIMS[args.grid]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel, tiny_cuda=args.checkpoint == "tcnn")
POOL = 96
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)}
          for t in range(POOL)]
torch.cuda.synchronize()


def digest(c, s):
    h = hashlib.sha256()
    if c is not None:
        h.update(c.cpu().numpy().tobytes())
        h.update(s.cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def run(depth):
    if args.world == 1:
        m = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=1 << 21, device="cuda:0", tsdf=True)
        m.inputs_resident = True
        vol = m.volume
        submit = lambda fr, nxt: m.fuse_and_decode_async(fr)                      # noqa: E731
    else:
        import torch.distributed as dist
        from bnv_fusion_amd import distributed as D
        if not dist.is_initialized():
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        be = D.HipShardBackend(np.array([dims] * 3), voxel, model, args.rank, args.world, capacity=1 << 21, device="cuda:0", tsdf=True,
                               n_slots=6)
        be.inputs_resident = True
        vol = be.volume
        ranks = torch.arange(args.world, dtype=torch.int32, device="cuda:0")
        pre = [None]

        def submit(fr, nxt):
            if pre[0] is not None:
                f, pre[0] = pre[0], None
            else:
                f = be.encode(fr)
            if nxt is not None and depth > 1:
                pre[0] = be.encode(nxt)
            cap = be.exchange_capacity(be.bound(f))
            send = be.upsert(f, cap, True)
            if cap:
                one = be.recv_buffer(args.world * send.numel())
                blocks = one.view(args.world, cap + 1, D.REC_WORDS)
                dist.all_gather_into_tensor(blocks[args.rank].reshape(-1), send)
                blocks[:] = blocks[args.rank].clone()
                blocks[:, 0, 1] = ranks
                be.install(f, one, cap)
            h = be.finish(f, be.decode(f), 0)

            class H:
                def result(self_):
                    return be.result(h)
            return H()
    out, pend = [], []
    with torch.no_grad():
        for i in range(args.frames):
            while len(pend) >= depth:
                out.append(digest(*pend.pop(0).result()))
            pend.append(submit(frames[i % POOL], frames[(i + 1) % POOL] if i + 1 < args.frames else None))
        while pend:
            out.append(digest(*pend.pop(0).result()))
    torch.cuda.synchronize()
    n = vol.num_rows()
    hv = hashlib.sha256()
    for t in (vol._row_coords[:n], vol._features[:n], vol._weights[:n]):
        hv.update(t.cpu().numpy().tobytes())
    return out, hv.hexdigest()[:16], n


t0 = time.time()
a, va, na = run(3)
t1 = time.time()
b, vb, nb = run(1)
bad = [i for i, (x, y) in enumerate(zip(a, b)) if x != y]
print(f"{args.frames} frames, world {args.world} rank {args.rank}, {args.checkpoint}: pipelined {t1 - t0:.1f} s, one at a time {time.time() - t1:.1f} s; "
      f"rows {na} / {nb}; volume digests {va} / {vb}; frames that differ: {len(bad)} {bad[:8]}")
sys.exit(0 if (not bad and va == vb and na == nb) else 1)
