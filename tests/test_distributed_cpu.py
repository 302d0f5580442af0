"""World-size-2 gloo tests (CPU) of the multi-GPU frame logic in bnv_fusion_amd/distributed.py: the spatial
sharding (ownership partition, the exchange bound every rank computes by itself, the ONE padded all-gather of
boundary records per frame, ghost rows) and the frame-parallel mode.  The compute of each rank is done by an
oracle-backed backend (test infrastructure); the union of the shards' SDF lattices must equal the single-process
oracle decode of the same frames."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN, WEIGHTS_FP32



class OracleShardBackend:
    """The shard protocol of bnv_fusion_amd.distributed.ShardedNeuralMap on the CPU oracle: same phases, same record
    layout (12 int32 words: x, y, z, weight bits, 8 feature bits; block = header record + capacity records)."""

    def __init__(self, dims, voxel, rank, world, ownership="hash"):
        from oracle import bnv_oracle as orc
        from bnv_fusion_amd import distributed as D
        self.orc, self.D = orc, D
        self.sd = orc.load_weights(WEIGHTS_FP32)
        self.vol = orc.OracleSparseVolume(8, voxel, dims, 8)
        self.rank, self.world, self.voxel = rank, world, voxel
        self.installed = 0
        # the ownership rule, as the host restatement of the device's owner table (distributed.OwnershipModel): every
        # rank feeds it the same replicated voxelisation and ends up with the same table -- no communication
        self.rule = D.OwnershipModel(ownership, world, self.vol.n_xyz.tolist())

    def encode(self, frame):
        o, v, D = self.orc, self.vol, self.D
        pts = frame["input_pts"]
        f, c, ids, g, n = o.encode_pointcloud(self.sd, pts, v.n_xyz, v.min_coords, v.max_coords, v.voxel_size)
        if f is None:          # no point inside the volume (local_point_fusion.py:101-102)
            return D.ShardFrame(grid_ids=torch.zeros((0, 3), dtype=torch.int64), counts=np.zeros(self.world, int),
                                n_avg=None)
        # the bound: touched BOUNDARY voxels per owner over ALL touched voxels (no min-points filter), replicated
        xyz = pts[0, :, :3]
        inb = ((xyz < (v.max_coords - v.voxel_size)) & (xyz > (v.min_coords + v.voxel_size))).all(-1)
        _, gid = o.get_relative_xyz(xyz[inb][None], v.min_coords, v.voxel_size)
        touched = torch.unique(gid.reshape(-1, 3).long(), dim=0).numpy()
        self.rule.frame(touched)                       # owners for the blocks this frame touches first (+ pins)
        bnd = self.rule.is_boundary(touched)
        counts = np.bincount(self.rule.owner(touched[bnd]), minlength=self.world)
        own = torch.from_numpy(self.rule.owner(g.numpy()) == self.rank)
        o.integrate(v, g[own], f[own], c[own])
        return D.ShardFrame(grid_ids=g[own], counts=counts, n_avg=n, feats=f[own], pcounts=c[own])

    def bound(self, fr):
        return int(fr.counts.max()) if self.world > 1 else 0

    def upsert(self, fr, capacity, decode=True):
        """(the oracle backend has already upserted in encode: this is the pack half)"""
        if capacity == 0:
            return None
        D, v = self.D, self.vol
        g = fr.grid_ids.numpy()
        send = g[self.rule.is_boundary(g)] if len(g) else g.reshape(0, 3)
        assert len(send) <= fr.counts[self.rank] <= capacity       # the bound bounds
        block = torch.zeros((capacity + 1, D.REC_WORDS), dtype=torch.int32)
        block[0, 0], block[0, 1] = len(send), self.rank
        if len(send):
            f, w, _ = v.query(torch.from_numpy(send))
            block[1: 1 + len(send), :3] = torch.from_numpy(send).int()
            block[1: 1 + len(send), 3] = w[:, 0].contiguous().view(torch.int32)
            block[1: 1 + len(send), 4:] = f.contiguous().view(torch.int32)
        return block.reshape(-1)

    def install(self, fr, blocks, capacity):
        D, v = self.D, self.vol
        blocks = blocks.reshape(self.world, capacity + 1, D.REC_WORDS)
        for r in range(self.world):
            if r == self.rank:
                continue
            n = int(blocks[r, 0, 0])
            assert int(blocks[r, 0, 1]) == r and int(blocks[r, 0, 2]) == 0 and n <= capacity
            rec = blocks[r, 1: 1 + n]
            keys = rec[:, :3].long()
            mine = torch.from_numpy(self.rule.adjacent_to(keys.numpy(), self.rank)) if n else torch.zeros(0, dtype=torch.bool)
            if mine.any():
                k = keys[mine]
                f_rec = rec[mine, 4:].contiguous().view(torch.float32)
                w_rec = rec[mine, 3:4].contiguous().view(torch.float32)
                v.insert(k, f_rec, w_rec, torch.zeros(len(k), 1))
                self.installed += len(k)
        return 0

    def decode(self, fr):
        o = self.orc
        if len(fr.grid_ids) == 0:
            return torch.zeros((0, 27))
        return self.vol.decode_pts(o.lattice_coords(fr.grid_ids.numpy()), self.sd, None, is_coords=True,
                                   query_tensor=False)[0, :, :, 0]

    def finish(self, fr, sdf, reserved):
        fr.sdf = sdf
        return fr

    def result(self, fr):
        return fr.grid_ids, fr.sdf

    def last_mlp_evals(self):
        return torch.zeros(1, dtype=torch.int32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, frames, dims, voxel, ret, ownership="hash"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bnv_fusion_amd.distributed import ShardedNeuralMap, all_gather_var
    nm = ShardedNeuralMap(dims, voxel, None, backend=OracleShardBackend(dims, voxel, rank, world, ownership))
    # an empty frame first (no point inside the volume): bound 0 on every rank -> no collective, (empty, empty) out
    far = torch.from_numpy(frames[0]).clone()
    far[..., :3] += 50.0
    e_owned, e_sdf = nm.fuse_and_decode({"input_pts": far})
    assert len(e_owned) == 0 and nm.exchanged_bytes == 0
    for fr in frames:
        owned, sdf = nm.fuse_and_decode({"input_pts": torch.from_numpy(fr)})
    assert nm.host_waits == len(frames) + 1 and nm.backend.installed > 0  # one host wait per frame; ghosts installed
    allc = all_gather_var(owned)
    alls = all_gather_var(sdf)
    ret[f"table{rank}"] = None if nm.backend.rule.table is None else nm.backend.rule.table.copy()
    v = nm.backend.vol
    v.to_tensor()                  # every row this rank holds (own + ghost): keys, features, weights
    ret[f"rows{rank}"] = (v.active_coordinates.numpy().copy(), v.features.numpy().copy(), v.weights.numpy().copy())
    if rank == 0:
        ret["coords"], ret["sdf"], ret["n0"] = allc.numpy(), alls.numpy(), len(owned)
        ret["bytes"] = nm.exchanged_bytes
    dist.destroy_process_group()


@pytest.mark.parametrize("ownership", ["hash", "region", "first_touch"])
def test_two_shards_equal_single_process(ownership):
    from oracle import bnv_oracle as orc
    from bnv_fusion_amd.distributed import OwnershipModel, touched_voxels, unflatten, voxel_owner
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    frames = list(z["frames"])      # 12 frames x 6000 points: weights reach min_pts
    dims, voxel = z["dims"], float(z["voxel_size"])
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), frames, dims, voxel, ret, ownership), nprocs=2, join=True)
        coords, sdf, n0 = ret["coords"], ret["sdf"], ret["n0"]
        tables = [ret["table0"], ret["table1"]]
        rows = [ret["rows0"], ret["rows1"]]
    # single-process reference
    sd = orc.load_weights(WEIGHTS_FP32)
    vol = orc.OracleSparseVolume(8, voxel, dims, 8)
    for fr in frames:
        f, c, _, g, _ = orc.encode_pointcloud(sd, torch.from_numpy(fr), vol.n_xyz, vol.min_coords, vol.max_coords, voxel)
        orc.integrate(vol, g, f, c)
    ref = vol.decode_pts(orc.lattice_coords(g.numpy()), sd, None, is_coords=True, query_tensor=False)[0, :, :, 0]
    # every row a rank holds -- its own and its ghost rows -- carries the single volume's values, bit for bit
    for keys, feats, weights in rows:
        f1, w1, _ = vol.query(torch.from_numpy(keys))
        assert np.array_equal(f1.numpy(), feats) and np.array_equal(w1.numpy(), weights) and len(keys) > 0
    # the shards partition the touched set exactly, by the ownership rule -- whose table both ranks derived alone, and
    # which is what a third party feeding the model the same frames derives too
    if ownership == "hash":
        own = voxel_owner(coords, 2)
    else:
        assert np.array_equal(tables[0], tables[1])
        rule = OwnershipModel(ownership, 2, vol.n_xyz.tolist())
        for fr in frames:
            ids, _ = touched_voxels(fr[0], vol.min_coords.numpy(), vol.max_coords.numpy(), voxel, vol.n_xyz.tolist())
            rule.frame(unflatten(ids, vol.n_xyz.tolist()))
        assert np.array_equal(rule.table, tables[0])
        own = rule.owner(coords)
    assert np.all(own[:n0] == 0) and np.all(own[n0:] == 1) and 0 < n0 < len(coords)
    order = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))
    assert np.array_equal(coords[order], g.numpy())
    assert np.abs(sdf[order] - ref.numpy()).max() < 2e-6
    assert np.array_equal(sdf[order] == np.float32(voxel), ref.numpy() == np.float32(voxel))
    assert (ref != voxel).float().mean() > 0.05          # the decode mask is live in this test


def test_boundary_predicates():
    """shard_is_boundary / shard_adjacent_to (host restatements of the device predicates): a voxel is a boundary
    voxel iff some OTHER rank is adjacent to it; interior voxels of a block are never boundary voxels; every voxel is
    adjacent to its owner."""
    from bnv_fusion_amd.distributed import shard_adjacent_to, shard_is_boundary, voxel_owner
    g = np.stack(np.meshgrid(np.arange(8, 40), np.arange(8, 40), np.arange(8, 24), indexing="ij"), -1).reshape(-1, 3)
    for world in (2, 4, 8):
        own = voxel_owner(g, world)
        bnd = shard_is_boundary(g, world)
        adj = np.stack([shard_adjacent_to(g, world, r) for r in range(world)], 1)
        assert adj[np.arange(len(g)), own].all()
        others = adj.copy()
        others[np.arange(len(g)), own] = False
        assert np.array_equal(bnd, others.any(1))
        interior = ((g & 7) > 0).all(1) & ((g & 7) < 7).all(1)
        assert not bnd[interior].any() and 0.2 < bnd.mean() < 0.8
    assert not shard_is_boundary(g, 1).any()


def test_owner_hash_is_balanced_and_blocked():
    from bnv_fusion_amd.distributed import voxel_owner
    g = np.stack(np.meshgrid(np.arange(64), np.arange(64), np.arange(64), indexing="ij"), -1).reshape(-1, 3)
    o = voxel_owner(g, 8)
    assert np.array_equal(voxel_owner(g, 1), np.zeros(len(g), dtype=np.int64))
    cnt = np.bincount(o, minlength=8)
    assert cnt.min() > 0.5 * cnt.mean() and cnt.max() < 1.6 * cnt.mean()
    blk = (g >> 3)
    key = (blk[:, 0] * 8 + blk[:, 1]) * 8 + blk[:, 2]
    for k in np.unique(key)[:20]:
        assert len(np.unique(o[key == k])) == 1           # whole 8^3 blocks share an owner


# ---------------------------------------------------------------------------------------------
# frame-parallel mode
# ---------------------------------------------------------------------------------------------
class OracleFrameBackend:
    """The frame-backend protocol of bnv_fusion_amd.distributed (headers + sized payloads) on top of the oracle."""

    def __init__(self, dims, voxel):
        from oracle import bnv_oracle as orc
        self.orc = orc
        self.sd = orc.load_weights(WEIGHTS_FP32)
        self.vol = orc.OracleSparseVolume(8, voxel, dims, 8)
        self.dev = torch.device("cpu")
        self.payload_rows = []

    def encode_frame(self, frame):
        from bnv_fusion_amd.distributed import EncodedFrame, header_counters
        v = self.vol
        f, c, _, g, n = self.orc.encode_pointcloud(self.sd, frame["input_pts"], v.n_xyz, v.min_coords, v.max_coords,
                                                   v.voxel_size)
        k = len(g)
        cap = k + 37                               # capacity-sized outputs; rows beyond n_out are don't-care: poison
        hdr = torch.zeros(8, dtype=torch.int64)
        counters = header_counters(hdr)
        counters[0], counters[2] = 1, k
        counters[3:4] = torch.tensor([float(n)]).view(torch.int32)
        grid_ids = torch.full((cap, 3), 0x7ff8dead, dtype=torch.int64)
        pcounts = torch.full((cap,), 0x7ff8dead, dtype=torch.int64)
        feats = torch.full((cap, 8), float("nan"))
        grid_ids[:k], pcounts[:k], feats[:k] = g, c.reshape(-1), f
        return EncodedFrame(hdr, grid_ids, pcounts, feats)

    def empty_frame(self):
        from bnv_fusion_amd.distributed import EncodedFrame
        return EncodedFrame(torch.zeros(8, dtype=torch.int64))

    def pack(self, enc, rows):
        from bnv_fusion_amd.distributed import payload_views, payload_words
        p = torch.full((payload_words(rows),), 0x7ff8dead, dtype=torch.int64)
        if enc.grid_ids is not None:
            g, c, f = payload_views(p, rows)
            m = min(rows, len(enc.pcounts))
            g[:m], c[:m], f[:m] = enc.grid_ids[:m], enc.pcounts[:m], enc.feats[:m]
        self.payload_rows.append(rows)
        return p

    def side(self, after_main):
        import contextlib
        return contextlib.nullcontext()

    def adopt(self, *tensors):
        pass

    def _valid(self, payload, rows, k):
        from bnv_fusion_amd.distributed import payload_views
        grid_ids, pcounts, feats = payload_views(payload, rows)
        return grid_ids[:k], pcounts[:k], feats[:k]

    def integrate_record(self, hdr, payload, rows, n_out, frame=None):
        from bnv_fusion_amd.distributed import header_counters
        assert int(header_counters(hdr)[2]) == n_out
        if n_out:
            g, c, f = self._valid(payload, rows, n_out)
            self.orc.integrate(self.vol, g, f, c.reshape(-1, 1))

    def integrate_records(self, hdr_all, out, rows, n_out, s0, s1):
        for s in range(s0, s1):
            self.integrate_record(hdr_all[s], None if out is None else out[s], rows, n_out[s])

    def integrate_tsdf(self, frames, n_valid=None):
        pass

    def decode_record(self, hdr, payload, rows, n_out):
        g, _, _ = self._valid(payload, rows, n_out)
        o = self.orc      # a sample of the voxels keeps the CPU suite fast; the exchange logic is what is tested
        return self.vol.decode_pts(o.lattice_coords(g.numpy()[::12]), self.sd, None, is_coords=True,
                                   query_tensor=False)[0, :, :, 0]

    def account(self, headers, n_rows_after=None):
        from bnv_fusion_amd.distributed import header_counters
        for h in headers:
            c = header_counters(h)
            if int(c[0]):
                self.vol.track_n_pts(float(c[3:4].view(torch.float32)[0]))

    def pinned(self, shape):
        return torch.empty(shape, dtype=torch.int64)

    def event(self):
        return None

    def slice_result(self, payload, rows, sdf, n_out):
        return self._valid(payload, rows, n_out)[0], sdf


def _fp_worker(rank, world, port, frames, dims, voxel, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bnv_fusion_amd.distributed import FrameParallelNeuralMap
    nm = FrameParallelNeuralMap(dims, voxel, None, backend=OracleFrameBackend(dims, voxel))
    outs = []
    fr = [{"input_pts": torch.from_numpy(f)} for f in frames]
    batches = [fr[t0: t0 + world] for t0 in range(0, len(fr), world)]   # the last batch is ragged (11 frames, world 2)
    # first 2 batches one by one, the rest through the pipelined stream (encode k+1 before integrate k)
    for i in range(2):
        c, sdf = nm.process_batch(batches[i])
        outs.append((i * world + rank, c.numpy(), sdf.numpy()))
    for i, h in enumerate(nm.process_stream(batches[2:]), start=2):
        c, sdf = h.result()
        if c is not None:
            outs.append((i * world + rank, c.numpy(), sdf.numpy()))
    nm.flush()
    ret[rank] = (outs, np.asarray(nm.backend.vol.n_pts_list), len(nm.backend.vol._keys), list(nm.backend.payload_rows))
    dist.destroy_process_group()


def test_frame_parallel_equals_single_process():
    from oracle import bnv_oracle as orc
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    frames = list(z["frames"])[:11]
    dims, voxel = z["dims"], float(z["voxel_size"])
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_fp_worker, args=(2, _free_port(), frames, dims, voxel, ret), nprocs=2, join=True)
        got = {t: (c, s) for r in (0, 1) for t, c, s in ret[r][0]}
        npts = [ret[r][1] for r in (0, 1)]
        nkeys = [ret[r][2] for r in (0, 1)]
        prow = [ret[r][3] for r in (0, 1)]
    sd = orc.load_weights(WEIGHTS_FP32)
    vol = orc.OracleSparseVolume(8, voxel, dims, 8)
    assert sorted(got) == list(range(11))            # every frame decoded exactly once
    # payloads are sized by the largest frame of the batch (from the exchanged headers), the same on both ranks
    assert prow[0] == prow[1] and len(prow[0]) == 6
    for b, rows in enumerate(prow[0]):
        biggest = max(len(got[t][0]) for t in range(2 * b, min(2 * b + 2, 11)))
        assert rows % 1024 == 0 and biggest <= rows < biggest + 1024
    for t, fr in enumerate(frames):
        f, c, _, g, n = orc.encode_pointcloud(sd, torch.from_numpy(fr), vol.n_xyz, vol.min_coords, vol.max_coords, voxel)
        vol.track_n_pts(n)
        orc.integrate(vol, g, f, c)
        ref = vol.decode_pts(orc.lattice_coords(g.numpy()[::12]), sd, None, is_coords=True,
                             query_tensor=False)[0, :, :, 0]
        assert np.array_equal(got[t][0], g.numpy())
        assert np.array_equal(got[t][1], ref.numpy())           # same ops in the same order: bit-identical
    assert nkeys[0] == nkeys[1] == len(vol._keys)               # replicated volumes stay in lock step
    assert np.allclose(npts[0], vol.n_pts_list) and np.allclose(npts[1], vol.n_pts_list)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without a launcher starts N ranks itself (a child torch.distributed.run, before
    any GPU call) and leaves with the child's exit code; a WORLD_SIZE that contradicts --gpus is refused."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-launch"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"dry_run_launch": True, "world": 2, "gpus": 2, "self_launched": True}
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-launch"],
                         env=dict(env, WORLD_SIZE="3", RANK="0"), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "does not match" in bad.stderr


@pytest.mark.parametrize("rule", ["first_touch", "region"])
def test_first_touch_tables_keep_the_exchange_invariant(rule):
    """distributed.OwnershipModel (the host restatement every sharded GPU test compares the device's owner table with):
    whenever a voxel is touched, every block of its block's 3x3x3 neighbourhood has an owner, and an owner once given
    never changes -- what the boundary / ghost-row predicates of the exchange rely on; the cumulative loads account for
    every touched voxel once.  A window drifting over a static surface goes out of balance under the region rule, which
    then hands new territory out by the greedy rule for good; a static view never trips that."""
    from bnv_fusion_amd.distributed import OWN_ASSIGNED, OWN_RANK, OWN_TOUCHED, OwnershipModel, _OFF27
    n = np.array([64, 64, 64])
    rng = np.random.default_rng(3)
    for drifting in (True, False):
        m = OwnershipModel(rule, 4, n)
        seen = set()
        weight = 0
        prev = m.table.copy()
        for t in range(40):
            c0 = np.array([10 + (1.1 * t if drifting else 0), 12 + (0.6 * t if drifting else 0)])
            xy = (rng.random((3000, 2)) * 26 + c0).astype(np.int64)
            z = (30 + 6 * np.sin(xy[:, 0] / 7.0) * np.cos(xy[:, 1] / 5.0)).astype(np.int64)
            vox = np.unique(np.stack([xy[:, 0], xy[:, 1], z], 1), axis=0)
            vox = vox[(vox >= 1).all(1) & (vox < 63).all(1)]
            m.frame(vox)
            t_now = m.table
            was = (prev & OWN_ASSIGNED) != 0
            assert np.array_equal(t_now[was] & OWN_RANK, prev[was] & OWN_RANK)          # owners never change
            blocks = np.unique(vox >> 3, axis=0)
            for d in _OFF27:
                e = blocks + d
                ok = ((e >= 0) & (e < m.nb)).all(1)
                assert (t_now[m._bidx(e[ok])] & OWN_ASSIGNED).all()                      # the neighbourhood has owners
            assert (t_now[m._bidx(blocks)] & OWN_TOUCHED).all()
            new = [tuple(b) for b in blocks.tolist() if tuple(b) not in seen]
            first = np.array([tuple(v >> 3) in set(new) for v in vox]) if new else np.zeros(len(vox), bool)
            weight += int(first.sum())
            seen.update(new)
            assert int(m.load.sum()) == weight                                             # every voxel of a new block once
            assert (m.owner(vox) >= 0).all()
            prev = t_now.copy()
        if rule == "region":
            assert m.interleave == drifting
        assert m.load.max() <= (1.6 if rule == "region" else 1.25) * m.load.mean()


def _contact_worker(rank, world, port, ret, same_identity):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bnv_fusion_amd.distributed import first_contact
    try:
        ret[rank] = first_contact(rank, world, "cpu", backend="gloo", timeout_s=30.0, records=64,
                                  identity=("box", 7) if same_identity else None)
    except RuntimeError as e:
        ret[rank] = str(e)
    dist.destroy_process_group()


def test_first_contact_reports_and_checks_the_rank_set():
    """bnv_fusion_amd.distributed.first_contact (bench.py --gpus N runs it before the timed region): every rank's
    identity and the backend's own rank count come back, five frame-shaped all-gathers complete inside their timeout
    and their data is checked; two ranks that claim the SAME device are refused with a message naming them; a world
    size that contradicts the launcher is refused."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_contact_worker, args=(2, _free_port(), ret, False), nprocs=2, join=True)
        for r in (0, 1):
            c = ret[r]
            assert isinstance(c, dict), c
            assert c["ranks_seen_by_backend"] == 2 and c["distinct_devices"] == 2
            assert [x["rank"] for x in c["ranks"]] == [0, 1] and c["ranks"][0]["pid"] != c["ranks"][1]["pid"]
            fc = c["first_contact"]
            assert fc["all_gathers"] == 5 and len(fc["ms_this_rank"]) == 5 and fc["data_checked"]
            assert fc["bytes_per_rank"] == 65 * 48
        ret2 = mgr.dict()
        mp.spawn(_contact_worker, args=(2, _free_port(), ret2, True), nprocs=2, join=True)
        assert all(isinstance(ret2[r], str) and "share a device" in ret2[r] and "[[0, 1]]" in ret2[r] for r in (0, 1))


def test_first_contact_refuses_a_wrong_world_size():
    import torch.distributed as dist
    from bnv_fusion_amd.distributed import first_contact
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        with pytest.raises(RuntimeError, match="reports 1 ranks, the launcher said 2"):
            first_contact(0, 2, "cpu", backend="gloo")
        c = first_contact(0, 1, "cpu", backend="gloo", n_gathers=2)
        assert c["ranks_seen_by_backend"] == 1 and c["rccl_version"] is None
    finally:
        dist.destroy_process_group()


def _contact_timeout_worker(rank, world, port, ret):
    import time
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bnv_fusion_amd.distributed import first_contact
    try:
        # rank 1 takes part in the identity exchange and then never shows up for the all-gather
        ret[rank] = first_contact(rank, world, "cpu", backend="gloo", timeout_s=1.5, records=8, n_gathers=1 if rank == 0 else 0)
        if rank == 1:
            time.sleep(4.0)
    except RuntimeError as e:
        ret[rank] = str(e)
    os._exit(0)          # (the process group holds a collective that will never complete: no orderly shutdown)


def test_first_contact_times_out_on_a_rank_that_never_arrives():
    """A collective that does not complete fails the rank that waits for it after first_contact's OWN timeout, with a
    message that names the round -- not after the process group's (minutes)."""
    import time
    with mp.Manager() as mgr:
        ret = mgr.dict()
        t0 = time.time()
        mp.spawn(_contact_timeout_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        assert time.time() - t0 < 60
        assert isinstance(ret[0], str) and "all-gather 1 of 1 did not complete within 1.5 s on rank 0" in ret[0], ret[0]
        assert isinstance(ret[1], dict)
