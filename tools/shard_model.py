"""CPU model (numpy) of the spatially sharded frame: which voxels a frame touches / emits, who owns them under an
ownership rule, and how many SDF-MLP table entries every rank of a world evaluates -- no GPU, no MLP.

    python tools/shard_model.py [--world 8] [--grid 256] [--scene pan|sweep] [--rule hash|greedy|region] [--block-log2 3]
                                [--warm 64] [--frames 32] [--axis 1]

What it reproduces (checked against the GPU: the single-volume figure of the bench frame, 1.754 M evaluations per launch
in profiles/r04_pmc_meta.json, comes out as 1.754 M here; per-rank figures are asserted against the real shards in
tests/test_gpu_multiprocess.py):

* voxelisation as csrc/encode.hip does it in fp32: bounds mask, (x - bmin) / v, floor / ceil corners, unique, counts;
  a voxel is EMITTED when >= min_pts pairs fall into it; its weight grows by min(count / 32, 1) per frame
  (local_point_fusion.py:653-673);
* the decode of an emitted voxel: 27 lattice points at {-.5, 0, .5}^3; a point is LIVE when all 8 corner voxels carry
  weight >= min_pts (sparse_volume.py:768-833); a live point reads 8 table entries (corner voxel, local offset);
  the table kernel evaluates every DISTINCT entry live points of the rank's own voxels read;
* ownership rules (distributed.py / csrc/encode.hip): block hash, round 4's greedy first touch + lattice pin, round 5's
  region-growing first touch.

Reported per evaluated frame and as means: evaluations per rank, their sum over the single-volume count (duplicated
work = the halo), max / mean load, boundary-record fraction, bytes a rank receives per frame."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bnv_fusion_amd import distributed as D          # noqa: E402  (host restatements only; no GPU is touched)
from bnv_fusion_amd import synthetic                 # noqa: E402


def world_range(dim, voxel):
    d = np.asarray([dim] * 3, dtype=np.float64)
    mn = -d / 2 - voxel
    mx = d / 2 + voxel
    n = np.ceil((mx - mn) / voxel).astype(int)
    return mn, mn + voxel * n, n


class Frames:
    """input_pts [N, 6] float32 per frame index, for the bench's pan or the room sweep (sequence.py)."""

    def __init__(self, scene, grid):
        self.scene, self.grid = scene, grid
        if scene == "sweep":
            from bnv_fusion_amd import sequence
            self.seq = sequence
            self.dim, self.voxel, self.scale = sequence.DIMS[grid]
        else:
            self.dim, self.voxel = synthetic.GRID_DIMS[grid]

    def pts(self, t):
        if self.scene == "pan":
            return synthetic.frame(t)[0]
        seq = self.seq
        K = seq.intrinsics()
        T = seq.sweep_pose(t, self.scale)
        d = seq.render_depth(T, K, 480, 640, self.scale).numpy()
        rng = np.random.default_rng(1000 * t)
        d = d + rng.normal(0.0, 0.002, size=d.shape)
        d = np.round(d * 1000.0).astype(np.uint16).astype(np.float64) / 1000.0
        return synthetic.depth_to_input_pts(d, K, T, max_depth=3.0).astype(np.float32)


def voxelise(pts, bmin32, bmax32, voxel32, n):
    """-> (sorted unique flat ids int64, counts) of the (point, corner) pairs, as k_mark / k_rank produce them."""
    return D.touched_voxels(pts, bmin32, bmax32, voxel32, n)


_L = np.array([[a, b, c] for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)], dtype=np.int64)   # 2 * offset


def lattice_entries(vox, weight_ok, n):
    """For emitted voxels ``vox`` [m, 3]: the table entries (corner voxel id * 27 + local-offset index) their LIVE
    lattice points read, concatenated (with repeats), and the number of live points.  ``weight_ok``: bool grid."""
    out = []
    live_pts = 0
    nyz = n[1] * n[2]
    for l in _L:                                     # lattice point p = v + l / 2
        # per axis: l = -1 -> corners (v - 1: local +.5 -> index 2, v: local -.5 -> index 0); l = 0 -> corner v, local 0
        # (floor == ceil: all 8 corners collapse onto fewer voxels); l = +1 -> (v: local +.5, v + 1: local -.5)
        opts = []
        for a in range(3):
            if l[a] == -1:
                opts.append(((-1, 2), (0, 0)))
            elif l[a] == 0:
                opts.append(((0, 1),))
            else:
                opts.append(((0, 2), (1, 0)))
        corners = [(ox, oy, oz) for ox in opts[0] for oy in opts[1] for oz in opts[2]]
        live = np.ones(len(vox), dtype=bool)
        cid = []
        for (dx, lx), (dy, ly), (dz, lz) in corners:
            c = vox + np.array([dx, dy, dz])
            inside = ((c >= 0) & (c < n)).all(1)
            flat = (np.clip(c[:, 0], 0, n[0] - 1) * n[1] + np.clip(c[:, 1], 0, n[1] - 1)) * n[2] + np.clip(c[:, 2], 0, n[2] - 1)
            live &= inside & weight_ok[flat]
            cid.append(flat * 27 + (lx * 9 + ly * 3 + lz))
        live_pts += int(live.sum())
        for e in cid:
            out.append(e[live])
    return (np.concatenate(out) if out else np.zeros(0, dtype=np.int64)), live_pts


def run(args):
    fr = Frames(args.scene, args.grid)
    voxel = fr.voxel
    mn, mx, n = world_range(fr.dim, voxel)
    bmin32 = mn.astype(np.float32)
    bmax32 = mx.astype(np.float32)
    voxel32 = np.float32(voxel)
    nvox = int(n.prod())
    weight = np.zeros(nvox, dtype=np.float32)
    W, s = args.world, args.block_log2
    rule = D.OwnershipModel(args.rule, W, n, s, axis=args.axis)
    rows = []
    t0 = args.start
    for t in range(t0, t0 + args.warm + args.frames * args.stride):
        ids, cnt = voxelise(fr.pts(t), bmin32, bmax32, voxel32, n)
        if len(ids) == 0:
            continue
        coords = np.stack([ids // (n[1] * n[2]), (ids // n[2]) % n[1], ids % n[2]], 1)
        rule.frame(coords)                             # owners for this frame's new blocks (+ pins)
        emit = cnt >= 8
        weight[ids[emit]] += np.minimum(cnt[emit] / np.float32(32.0), np.float32(1.0)).astype(np.float32)
        if t < t0 + args.warm or (t - t0 - args.warm) % args.stride:
            continue
        ok = weight >= 8.0
        ev = coords[emit]
        own = rule.owner(ev)
        ent, live_pts = lattice_entries(ev, ok, n)
        single = len(np.unique(ent))
        per = []
        for r in range(W):
            e_r, _ = lattice_entries(ev[own == r], ok, n)
            per.append(len(np.unique(e_r)))
        per = np.array(per, dtype=np.float64)
        bnd = rule.is_boundary(ev)
        adj = np.array([int(rule.adjacent_to(ev[bnd & (own != r)], r).sum()) for r in range(W)])
        town = rule.owner(coords)
        pairs = np.array([cnt[town == r].sum() for r in range(W)], dtype=np.float64)
        rows.append(dict(t=t, touched=len(ids), emitted=int(emit.sum()), live=live_pts, single=single, per=per,
                         boundary=float(bnd.mean()), sent=np.array([int((bnd & (own == r)).sum()) for r in range(W)]),
                         ghosts=adj, pairs=pairs))
        if args.verbose:
            print(f"frame {t}: emitted {int(emit.sum())}, single {single}, per rank max {per.max():.0f} mean {per.mean():.0f} "
                  f"sum/single {per.sum() / single:.3f}, boundary {bnd.mean():.3f}", flush=True)
    if not rows:
        print("no frame with points")
        return None
    single = np.mean([r["single"] for r in rows])
    per = np.mean([r["per"] for r in rows], 0)
    worst = np.mean([r["per"].max() for r in rows])
    pairs = np.mean([r["pairs"] for r in rows], 0)
    sent = np.mean([r["sent"] for r in rows], 0)
    cap = np.mean([-(-r["sent"].max() // 512) * 512 for r in rows])
    out = dict(single=single, per=per, sum_over_single=per.sum() / single, slowest_over_ideal=worst / (single / W),
               max_over_mean=np.mean([r["per"].max() / r["per"].mean() for r in rows]),
               pairs_max_over_mean=np.mean([r["pairs"].max() / r["pairs"].mean() for r in rows]),
               boundary=np.mean([r["boundary"] for r in rows]), sent=sent, recv_mb=W * (cap + 1) * 48 / 1e6,
               ghosts=np.mean([r["ghosts"] for r in rows], 0), emitted=np.mean([r["emitted"] for r in rows]),
               frames=len(rows), worst=worst)
    print(f"{args.scene} {args.grid}^3, world {W}, rule {args.rule}, blocks {1 << s}^3" +
          (f", bands stacked along axis {args.axis}" if args.rule == "region" else "") +
          f": {len(rows)} frames behind {args.warm} warm-up frames")
    print(f"  single volume: {single:,.0f} evaluations per frame, {out['emitted']:,.0f} emitted voxels")
    print(f"  evaluations per rank (mean over frames): " + " ".join(f"{v:,.0f}" for v in per))
    print(f"  sum over ranks / single = {out['sum_over_single']:.3f}   max / mean per frame = {out['max_over_mean']:.3f}   "
          f"slowest rank / (single / world) = {out['slowest_over_ideal']:.3f}  ({worst:,.0f} evaluations)")
    print(f"  pairs max / mean = {out['pairs_max_over_mean']:.3f}   boundary records / emitted voxels = {out['boundary']:.3f}   "
          f"records sent per rank (mean) {sent.mean():,.0f}, installed as ghosts per rank {out['ghosts'].mean():,.0f}, "
          f"all-gather {out['recv_mb']:.2f} MB per rank and frame")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--scene", default="pan", choices=["pan", "sweep"])
    ap.add_argument("--rule", default="region", choices=["hash", "greedy", "region"])
    ap.add_argument("--block-log2", type=int, default=3)
    ap.add_argument("--axis", type=int, default=1, help="region rule: the axis the first frame's bands are stacked along")
    ap.add_argument("--warm", type=int, default=64)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--stride", type=int, default=1, help="evaluate every stride-th frame behind the warm-up")
    ap.add_argument("--verbose", action="store_true")
    run(ap.parse_args())
