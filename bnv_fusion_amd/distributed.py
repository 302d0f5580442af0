"""Multi-GPU local fusion + decode, one process per GPU, RCCL over xGMI (SURVEY.md section 8e -- new design; the
reference is single-GPU).  Two decompositions, both producing exactly the single-GPU outputs:

SPATIAL SHARDING (the north-star decomposition; ``ShardedNeuralMap``)
  Ownership   owner(voxel) = mix64(block coordinate) % world, blocks of 8^3 voxels (same function in
              csrc/bnv_common.hpp: voxel_owner).  A (point, corner) pair contributes to exactly one voxel, so every
              rank voxelises the whole frame (cheap, replicated) but runs the point encoder only on pairs whose voxel
              it owns and upserts only its own voxels: per-voxel sums are complete locally, NO reduction collective.
  Exchange    the decode of a voxel reads the rows of its 3x3x3 neighbourhood, some owned elsewhere: GHOST rows.
              A row changes only when a frame's encode emits its voxel, so per frame every rank sends the rows it has
              just updated that are BOUNDARY voxels (a voxel of their 3x3x3 neighbourhood has another owner: a function
              of the coordinates alone) -- 48-byte records {key, weight, 8 features} -- in ONE all-gather; every rank
              installs the records adjacent to voxels it owns.  Then its local volume (own + ghost rows) decodes the
              voxels it owns exactly as the single-GPU volume would.  Outputs stay sharded.
  Sizes       the all-gather is padded to the largest block.  Its size comes from a bound every rank computes for
              ALL ranks without talking to anybody: the voxelisation is replicated, so the number of touched boundary
              voxels each rank owns is known on every GPU right after the frame's voxels are ranked, BEFORE the
              encoder runs (bnv_encode_begin).  The host reads those few integers while the encoder runs: the one
              host wait of a frame.  The real record counts travel in the blocks' headers and stay on the device.
  Bytes       per frame and rank ~0.58 x U'/world records of 48 B at 8^3 blocks (U' ~ 1e5 emitted voxels at
              640x480): ~2.8 MB gathered per frame in total, ~0.35 MB sent per rank at world 8.

FRAME-PARALLEL (throughput of one frame stream; ``FrameParallelNeuralMap``, below)

The frame logic talks to a backend; ``HipShardBackend`` / ``HipFrameBackend`` are the product backends (HIP kernels).
tests/test_distributed_cpu.py drives the same logic over gloo with CPU backends of its own.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

BLOCK_LOG2 = 3          # default block edge (log2 voxels); HipShardBackend(block_log2=) / BNV_SHARD_BLOCK_LOG2 override it
# ownership rule of HipShardBackend when the caller names none (BNV_SHARD_OWNERSHIP overrides):
#   "region"       first touch, contiguous regions (include/bnv_fusion.h: BNV_SHARD_RULE_REGION): ~1.03-1.07 x the
#                  single-GPU decode work in total, max / mean load 1.04 while the view stays put or moves across the
#                  bands (the benchmark's pan) -- but 1.4-1.5 for a camera that sweeps a room;
#   "first_touch"  first touch, fine interleave (round 4; BNV_SHARD_RULE_GREEDY): max / mean 1.01-1.06 in ANY view, ~1.2 x
#                  the work (1.11 with block_log2 = 4): the rule for moving cameras;
#   "hash"         mix64(block) % world: no table, max / mean 1.11-1.16.
# tools/shard_model.py prices the three on the CPU (profiles/r05_shard_model.txt).
DEFAULT_OWNERSHIP = "region"
DEFAULT_AXIS = 1        # region rule: the first frame's bands are stacked along y (the vertical of the datasets' cameras)
RULES = {"first_touch": 0, "region": 1}
REC_WORDS = 12          # BNV_SHARD_RECORD_BYTES / 4: {x, y, z, weight bits, 8 feature bits}
REC_QUANTUM = 512       # the per-rank block capacity is rounded up to this many records


def mix64(k):
    k = np.asarray(k, dtype=np.uint64).copy()
    with np.errstate(over="ignore"):
        k ^= k >> np.uint64(33)
        k *= np.uint64(0xff51afd7ed558ccd)
        k ^= k >> np.uint64(33)
        k *= np.uint64(0xc4ceb9fe1a85ec53)
        k ^= k >> np.uint64(33)
    return (k & np.uint64(0xFFFFFFFF)).astype(np.uint64)


def lattice_owner(blocks, world):
    """The rule that pins blocks nobody has touched yet (csrc/bnv_common.hpp: shard_lattice_owner)."""
    b = np.asarray(blocks, dtype=np.int64)
    return (b[:, 0] + 5 * b[:, 1] + 7 * b[:, 2]) % world


def voxel_owner(coords, world, block_log2=BLOCK_LOG2, table=None, n_xyz=None):
    """Host restatement of csrc/bnv_common.hpp voxel_owner: coords [n, 3] int -> rank [n].  ``table`` (with ``n_xyz``):
    the first-touch owner table of a shard (HipShardBackend.owner_table()): -1 outside the grid / without an owner."""
    c = np.asarray(coords, dtype=np.int64)
    if world <= 1:
        return np.zeros(len(c), dtype=np.int64)
    if table is not None:
        n = np.asarray(n_xyz, dtype=np.int64)
        nb = (n + (1 << block_log2) - 1) >> block_log2
        inside = ((c >= 0) & (c < n)).all(1)
        b = np.where(inside[:, None], c, 0) >> block_log2
        t = np.asarray(table).reshape(-1)[(b[:, 0] * nb[1] + b[:, 1]) * nb[2] + b[:, 2]].astype(np.int64)
        return np.where(inside & ((t & 0x40) != 0), t & 0x3f, -1)
    b = (c >> block_log2).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    key = (b[:, 0] << np.uint64(42)) | (b[:, 1] << np.uint64(21)) | b[:, 2]
    return (mix64(key) % np.uint64(world)).astype(np.int64)


_OFF27 = np.array([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)], dtype=np.int64)


def shard_is_boundary(coords, world, block_log2=BLOCK_LOG2, table=None, n_xyz=None):
    """Host restatement of csrc/bnv_common.hpp shard_is_boundary: does a voxel of the 3x3x3 neighbourhood belong to
    another rank?  coords [n, 3] -> bool [n]."""
    c = np.asarray(coords, dtype=np.int64).reshape(-1, 3)
    me = voxel_owner(c, world, block_log2, table, n_xyz)
    out = np.zeros(len(c), dtype=bool)
    for d in _OFF27:
        o = voxel_owner(c + d, world, block_log2, table, n_xyz)
        if table is None:
            out |= o != me
        else:       # first-touch rule: neighbours outside the grid hold no voxel and do not count
            inside = ((c + d >= 0) & (c + d < np.asarray(n_xyz, dtype=np.int64))).all(1)
            out |= inside & (o != me)
    return out


def shard_adjacent_to(coords, world, rank, block_log2=BLOCK_LOG2, table=None, n_xyz=None):
    """Host restatement of shard_adjacent_to: does ``rank`` own a voxel of the 3x3x3 neighbourhood (itself included)?"""
    c = np.asarray(coords, dtype=np.int64).reshape(-1, 3)
    out = np.zeros(len(c), dtype=bool)
    for d in _OFF27:
        out |= voxel_owner(c + d, world, block_log2, table, n_xyz) == rank
    return out


def touched_voxels(pts, bound_min, bound_max, voxel_size, n_xyz):
    """Host restatement (numpy, fp32 like csrc/encode.hip: k_mark + k_rank) of a frame's voxelisation: input_pts
    [N, >= 3] -> (ascending flat ids int64 [U], pair counts [U]) of the voxels its (point, corner) pairs fall into."""
    n = np.asarray(n_xyz, dtype=np.int64)
    bmin = np.asarray(bound_min, dtype=np.float32).reshape(3)
    v = np.float32(voxel_size)
    lo = (bmin + v).astype(np.float32)
    hi = (np.asarray(bound_max, dtype=np.float32).reshape(3) - v).astype(np.float32)
    x = np.asarray(pts)[:, :3].astype(np.float32)
    with np.errstate(invalid="ignore"):
        x = x[((x < hi) & (x > lo)).all(1)]                  # strict, one-voxel margin; NaN rows fail
    xn = (x - bmin) / v                                      # two IEEE fp32 roundings, as voxel_coord()
    f = np.floor(xn).astype(np.int64)
    c = np.ceil(xn).astype(np.int64)
    ids = []
    for bits in range(8):
        cx = np.where(bits & 1, c[:, 0], f[:, 0])
        cy = np.where(bits & 2, c[:, 1], f[:, 1])
        cz = np.where(bits & 4, c[:, 2], f[:, 2])
        ids.append((cx * n[1] + cy) * n[2] + cz)
    if not len(x):
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    return np.unique(np.concatenate(ids), return_counts=True)


def unflatten(ids, n_xyz):
    n = np.asarray(n_xyz, dtype=np.int64)
    ids = np.asarray(ids, dtype=np.int64)
    return np.stack([ids // (n[1] * n[2]), (ids // n[2]) % n[1], ids % n[2]], 1)


OWN_RANK, OWN_ASSIGNED, OWN_TOUCHED = 0x3f, 0x40, 0x80        # bits of an owner-table byte (csrc/bnv_common.hpp)


def walk_key(b, nb, axis):
    """Walk order of a frame's new blocks under the region rule: block coordinates [n, 3] -> sort key.  Bands are
    stacked along ``axis``: that coordinate is the most significant, the other two follow in x, y, z order
    (csrc/bnv_common.hpp: shard_walk_key)."""
    a1, a2 = [a for a in range(3) if a != axis]
    return (b[:, axis] * nb[a1] + b[:, a1]) * nb[a2] + b[:, a2]


class OwnershipModel:
    """Host restatement of the ownership rules of a spatially sharded volume (csrc/encode.hip: k_rank, k_shard_assign;
    csrc/bnv_common.hpp: voxel_owner, shard_is_boundary, shard_adjacent_to): fed the touched voxels of every frame in
    order, it holds the owner table every rank's device table must equal -- the specification the GPU tests compare
    with, and what tools/shard_model.py prices rules with.

    ``hash``    owner = mix64(block) % world.
    ``greedy``  round 4's first touch: a frame's new blocks, ascending, each to the rank with the least cumulative load;
                untouched neighbour blocks pinned to the lattice rule (bx + 5 by + 7 bz) % world.
    ``region``  round 5's first touch (the default): contiguous regions.  cur[r] = the voxels THIS frame touches in
                blocks rank r owns.  A new block without an owner, in walk order (bands stacked along ``axis``), goes to
                the least-loaded owner among the 26 blocks around it (``_OFF27``, as the device kernel) that is not full (cur * world < touched voxels of
                the frame), else to the least-loaded rank of all; a new block pinned earlier keeps its owner.  Then the
                untouched neighbour blocks of the new blocks are pinned to the owner of the first new block (walk order)
                that reaches them -- regions grow outwards -- unless that rank is overloaded
                (cur * world * 8 > 9 * touched), then to the least-loaded rank: it starts a new region there.
                Regions balance the load only while the view stays put: the first frame whose most loaded rank carries
                more than 1.3 x its share switches the table to the greedy rule for all territory to come
                (``interleave``, sticky): a camera that sweeps a room gets the fine interleave."""

    def __init__(self, rule, world, n_xyz, block_log2=BLOCK_LOG2, axis=1, pin_num=9, pin_den=8):
        assert rule in ("hash", "greedy", "region", "first_touch")
        self.pin_num, self.pin_den = int(pin_num), int(pin_den)
        self.recv = -1
        self.interleave = False      # region rule: True once a frame's load was out of balance (a sweeping camera)
        self.rule = "greedy" if rule == "first_touch" else rule
        self.world, self.s, self.axis = int(world), int(block_log2), int(axis)
        self.n = np.asarray(n_xyz, dtype=np.int64)
        self.nb = (self.n + (1 << self.s) - 1) >> self.s
        self.table = None if self.rule == "hash" else np.zeros(int(self.nb.prod()), dtype=np.uint8)
        self.load = np.zeros(self.world, dtype=np.uint64)      # cumulative: weights of the blocks at first touch
        self.cur = np.zeros(self.world, dtype=np.int64)        # region rule: the last frame's load per rank

    # ---- predicates with the table as it stands ---------------------------------------------------
    def owner(self, coords):
        return voxel_owner(coords, self.world, self.s, self.table, self.n)

    def is_boundary(self, coords):
        return shard_is_boundary(coords, self.world, self.s, self.table, self.n)

    def adjacent_to(self, coords, rank):
        c = np.asarray(coords, dtype=np.int64).reshape(-1, 3)
        if len(c) == 0:
            return np.zeros(0, dtype=bool)
        return shard_adjacent_to(c, self.world, rank, self.s, self.table, self.n)

    def _bidx(self, b):
        return (b[..., 0] * self.nb[1] + b[..., 1]) * self.nb[2] + b[..., 2]

    def _bcoord(self, i):
        i = np.asarray(i, dtype=np.int64)
        return np.stack([i // (self.nb[1] * self.nb[2]), (i // self.nb[2]) % self.nb[1], i % self.nb[2]], -1)

    # ---- one frame --------------------------------------------------------------------------------
    def frame(self, touched):
        """``touched``: [U, 3] voxel coordinates this frame touches (every voxel once)."""
        if self.table is None or self.world <= 1:
            return
        t = np.asarray(touched, dtype=np.int64).reshape(-1, 3)
        if len(t) == 0:
            return
        T = self.table
        bi, w = np.unique(self._bidx(t >> self.s), return_counts=True)
        new = (T[bi] & OWN_TOUCHED) == 0
        if self.rule == "region":
            # contiguous regions keep the load level only while the view stays put: once the most loaded rank carries
            # more than 1.3 x its share of a frame's voxels, new territory goes by the greedy rule for good
            cur = np.zeros(self.world, dtype=np.int64)
            asg = (T[bi] & OWN_ASSIGNED) != 0
            np.add.at(cur, (T[bi[asg]] & OWN_RANK).astype(np.int64), w[asg])
            if cur.max() * self.world * 10 > 13 * len(t):
                self.interleave = True
        if self.rule == "greedy" or self.interleave:
            for b, wt in zip(bi[new], w[new]):                       # ascending block index
                if T[b] & OWN_ASSIGNED:
                    r = int(T[b] & OWN_RANK)
                else:
                    r = int(np.argmin(self.load))                    # (first minimum: lowest rank on ties)
                self.load[r] += np.uint64(wt)
                T[b] = r | OWN_ASSIGNED | OWN_TOUCHED
            for b in bi[new]:
                for d in _OFF27:
                    e = self._bcoord(b) + d
                    if (e < 0).any() or (e >= self.nb).any():
                        continue
                    k = int(self._bidx(e))
                    if not T[k] & OWN_ASSIGNED:
                        T[k] = int(lattice_owner(e[None], self.world)[0]) | OWN_ASSIGNED
            return
        # ---- region rule
        n_touched = len(t)
        W = self.world
        cur = np.zeros(W, dtype=np.int64)
        asg = (T[bi] & OWN_ASSIGNED) != 0
        np.add.at(cur, (T[bi[asg]] & OWN_RANK).astype(np.int64), w[asg])
        nbi, nw = bi[new], w[new]
        self.cur = cur
        if len(nbi) == 0:
            return                  # (a frame without a new block changes nothing: the receiver keeps its role)
        order = np.argsort(walk_key(self._bcoord(nbi), self.nb, self.axis), kind="stable")
        nbi, nw = nbi[order], nw[order]

        def least(c):
            return int(np.lexsort((np.arange(W), c))[0])

        # the RECEIVER: the rank new territory goes to when adjacency does not decide.  A rank is FULL -- no more new
        # blocks for it in this frame -- when it carries its share of the frame's voxels; a rank is OVERLOADED -- no pins
        # for it -- when it carries more than pin_num / pin_den of its share.
        def full(c):
            return cur[c] * W >= n_touched

        def overloaded(c):
            return cur[c] * W * self.pin_den > self.pin_num * n_touched

        recv = self.recv
        if recv < 0 or full(recv):
            recv = least(cur)
        for b, wt in zip(nbi, nw):
            if T[b] & OWN_ASSIGNED:
                r = int(T[b] & OWN_RANK)                             # pinned earlier: cur already counts its voxels
            else:
                bc = self._bcoord(b)
                best = -1
                for d in _OFF27:
                    e = bc + d
                    if (e < 0).any() or (e >= self.nb).any():
                        continue
                    v = T[int(self._bidx(e))]
                    if not v & OWN_ASSIGNED:
                        continue
                    c = int(v & OWN_RANK)
                    if full(c):
                        continue
                    if best < 0 or (cur[c], c) < (cur[best], best):
                        best = c
                if best >= 0:
                    r = best
                else:
                    if full(recv):
                        recv = least(cur)
                    r = recv
                cur[r] += wt
            self.load[r] += np.uint64(wt)
            T[b] = r | OWN_ASSIGNED | OWN_TOUCHED
        if overloaded(recv):
            recv = least(cur)
        for b in nbi:
            bc = self._bcoord(b)
            r = int(T[b] & OWN_RANK)
            if overloaded(r):
                r = recv
            for d in _OFF27:
                e = bc + d
                if (e < 0).any() or (e >= self.nb).any():
                    continue
                k = int(self._bidx(e))
                if not T[k] & OWN_ASSIGNED:
                    T[k] = r | OWN_ASSIGNED
        self.recv = recv
        self.cur = cur


def all_gather_var(t, group=None):
    """All-gather of tensors whose first dimension differs per rank -> concatenation in rank order (host-synchronous;
    used by tests and tools to collect sharded OUTPUTS, not on the per-frame path)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)


def first_contact(rank, world, device, group=None, backend=None, n_gathers=5, timeout_s=60.0, records=512,
                  identity=None):
    """First contact of a freshly made process group, BEFORE anything is timed (bench.py --gpus N; RCCL with more than
    one rank has never run on the development boxes, so the first minutes on a node must fail loudly, not hang):

    * every rank reports {rank, host, pid, device, device name, PCI bus id, uuid} (all_gather_object) and the rank
      count the backend itself reports; a world size that differs from ``world`` raises;
    * with a GPU backend every rank must sit on a DISTINCT physical device -- two ranks on one GPU (a launcher that
      did not set LOCAL_RANK, a container that exposes one device) would run, slowly, and report a meaningless scaling
      curve: raises, naming the ranks that clash (``identity`` overrides what is compared: tests);
    * ``n_gathers`` all-gathers shaped like a frame's exchange ((1 + records) 48-byte records per rank), each block
      tagged with its sender and the round, each waited for with ITS OWN timeout (``timeout_s``), and the landed
      blocks checked on the host: a collective that does not complete, or lands the wrong data, raises a RuntimeError
      that names the round -- the caller turns it into a non-zero exit (never a re-exec: bench.py's ranks are children
      of the launcher).

    -> a dict for the bench line's ``distributed`` entry.  CPU backends (gloo) run the same checks on host tensors."""
    import os
    import socket
    import time
    import torch.distributed as dist
    dev = torch.device(device)
    on_gpu = dev.type == "cuda"
    me = {"rank": int(rank), "host": socket.gethostname(), "pid": os.getpid(), "device": str(dev)}
    if on_gpu:
        pr = torch.cuda.get_device_properties(dev)
        bus = "%04x:%02x:%02x" % (int(getattr(pr, "pci_domain_id", 0)), int(getattr(pr, "pci_bus_id", -1)),
                                  int(getattr(pr, "pci_device_id", 0)))
        me.update(device_name=pr.name, pci_bus_id=bus, uuid=str(getattr(pr, "uuid", "")),
                  compute_units=int(pr.multi_processor_count))
    me["identity"] = identity if identity is not None else (
        (me["host"], me.get("pci_bus_id"), me.get("uuid")) if on_gpu else (me["host"], me["pid"]))
    seen = dist.get_world_size(group)
    if seen != int(world):
        raise RuntimeError(f"first contact: the process group reports {seen} ranks, the launcher said {world}")
    ranks = [None] * seen
    dist.all_gather_object(ranks, me, group=group)
    if sorted(r["rank"] for r in ranks) != list(range(seen)):
        raise RuntimeError(f"first contact: ranks are not 0..{seen - 1}: {[r['rank'] for r in ranks]}")
    ranks.sort(key=lambda r: r["rank"])
    by_id = {}
    for r in ranks:
        by_id.setdefault(tuple(r["identity"]) if isinstance(r["identity"], (list, tuple)) else r["identity"], []).append(r["rank"])
    clash = [v for v in by_id.values() if len(v) > 1]
    if clash and (on_gpu or identity is not None):
        raise RuntimeError(f"first contact: ranks {clash} share a device ({[ranks[v[0]]['identity'] for v in clash]}): "
                           "one process per GPU -- check LOCAL_RANK / the visible devices")
    words = (1 + int(records)) * REC_WORDS
    send = torch.zeros(words, dtype=torch.int32, device=dev)
    recv = torch.empty(seen * words, dtype=torch.int32, device=dev)
    ms = []
    for k in range(int(n_gathers)):
        send.fill_(int(rank) * 1000 + k)
        send[0], send[1] = int(records), int(rank)              # a block's header: {count, sender}
        recv.fill_(-1)
        if on_gpu:
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        work = dist.all_gather_into_tensor(recv, send, group=group, async_op=True)
        # the HOST polls with its own deadline (with RCCL, work.wait() only orders the current stream behind the
        # collective and a device synchronisation on a wedged collective would never return)
        t_end = t0 + float(timeout_s)
        try:
            while not work.is_completed():
                if time.perf_counter() > t_end:
                    raise RuntimeError(f"first contact: all-gather {k + 1} of {n_gathers} did not complete within "
                                       f"{timeout_s} s on rank {rank} ({seen} ranks, {words * 4} bytes per rank)")
                time.sleep(0.0005)
            work.wait()
            if on_gpu:
                torch.cuda.synchronize(dev)
        except RuntimeError:
            raise
        except Exception as e:       # noqa: BLE001 -- the backend's own timeout / abort
            raise RuntimeError(f"first contact: all-gather {k + 1} of {n_gathers} failed on rank {rank} "
                               f"({seen} ranks, {words * 4} bytes per rank): {type(e).__name__}: {e}") from e
        ms.append(1e3 * (time.perf_counter() - t0))
        got = recv.view(seen, words).cpu()
        for r in range(seen):
            if int(got[r, 1]) != r or int(got[r, 0]) != int(records) or not bool((got[r, 2:] == r * 1000 + k).all()):
                raise RuntimeError(f"first contact: all-gather {k + 1} landed wrong data for sender {r} on rank {rank} "
                                   f"(header {got[r, :2].tolist()}, payload {int(got[r, 2])})")
    ver = None
    if on_gpu and (backend or "nccl") == "nccl":
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:           # noqa: BLE001
            ver = None
    for r in ranks:
        r.pop("identity", None)
    return {"ranks_seen_by_backend": seen, "distinct_devices": len(by_id), "ranks": ranks, "rccl_version": ver,
            "first_contact": {"all_gathers": int(n_gathers), "bytes_per_rank": words * 4, "timeout_s": float(timeout_s),
                              "ms_this_rank": ms, "data_checked": True}}


class ShardFrame:
    """One frame on its way through a shard backend (everything a later phase needs)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class HipShardBackend:
    """One shard of the volume on one GPU: the frame chain of csrc/pipeline.hip (FramePipe) with persistent per-slot
    buffers -- the phases below enqueue launches and nothing else; no tensor, event or pinned buffer is made per
    frame.  Host waits: ``bound`` (the frame's ONE wait on the data path; it returns while the main stream still
    holds the previous frame) and ``result``."""

    def __init__(self, dimensions, voxel_size, pointnet, rank, world, min_pts_in_grid=8, capacity=1 << 20,
                 device="cuda:0", tsdf=False, max_depth=3.0, n_slots=4, ownership=None, block_log2=None, axis=None,
                 encoder_workgroups=None):
        from .sparse_volume import SparseVolume, make_grid
        import os
        ownership = ownership or os.environ.get("BNV_SHARD_OWNERSHIP", DEFAULT_OWNERSHIP)
        if ownership not in ("hash", "first_touch", "region"):
            raise ValueError(f"unknown ownership rule {ownership!r}")
        self.ownership = ownership
        self.block_log2 = int(block_log2 if block_log2 is not None else os.environ.get("BNV_SHARD_BLOCK_LOG2", BLOCK_LOG2))
        if not 1 <= self.block_log2 <= 6:
            raise ValueError(f"block_log2 {self.block_log2} outside 1..6")
        self.axis = int(axis if axis is not None else os.environ.get("BNV_SHARD_AXIS", DEFAULT_AXIS))
        self.pointnet = pointnet
        self.rank, self.world = rank, world
        self.volume = SparseVolume(8, voxel_size, dimensions, min_pts_in_grid, capacity=capacity, device=device)
        v = self.volume
        # first-touch ownership (include/bnv_fusion.h: bnv_grid_t.shard_state): the owner table lives on the device,
        # is updated by every frame's encode and holds the same content on every rank
        self._owner_state = None
        if ownership != "hash" and world > 1:
            n_arr = (C.c_int32 * 3)(*v._n_xyz_host)
            nbytes = int(v._lib.bnv_shard_state_bytes(n_arr, self.block_log2))
            self._owner_state = torch.zeros(nbytes, dtype=torch.uint8, device=v._dev)
            _lib.check(v._lib.bnv_shard_state_configure(_lib.ptr(self._owner_state), RULES[ownership], self.axis,
                                                        _lib.stream_ptr()), "bnv_shard_state_configure")
        self.shard = (rank, world, self.block_log2) + ((self._owner_state.data_ptr(),) if self._owner_state is not None else ())
        pointnet.shard = self.shard
        v.shard = self.shard
        v._grid = make_grid(v._n_xyz_host, v.min_coords, v.max_coords, voxel_size, min_pts_in_grid, v.shard)
        self.dev = v._dev
        self.max_depth = max_depth
        self.sdf_delta = None
        self.tsdf_vol = None
        if tsdf:            # the TSDF side volume (0.025 m, dense) is small: every rank keeps the whole of it
            from .sparse_volume import get_world_range
            from .tsdf import TSDFVolume
            mn, mx, _ = get_world_range(dimensions, 0.025)
            self.tsdf_vol = TSDFVolume(np.stack([mn, mx], 1), 0.025, device=device)
        self._lib = v._lib
        self.n_slots = n_slots
        self.inputs_resident = False      # True: frames are complete in device memory when they are passed in
        self.copy_results = True          # False: result() returns views into the slot buffers (valid for n_slots - 1 more frames)
        self.pipe = None
        self._recv = None
        self._pipe_kw = {} if encoder_workgroups is None else {"encoder_workgroups": encoder_workgroups}
        self._last_evals = 0
        self.last_owned_pairs = 0

    def _pipe_for(self, frame):
        n = int(frame["input_pts"].shape[1]) if "input_pts" in frame else int(frame["depth"].shape[-2] * frame["depth"].shape[-1])
        if self.pipe is None or self.pipe.max_points < n:
            from .pipeline import FramePipe
            assert self.pipe is None or not any(self.pipe._busy), "a larger frame arrived while frames are in flight"
            self.pipe = FramePipe(self.volume, self.pointnet, n, n_slots=self.n_slots, tsdf_vol=self.tsdf_vol,
                                  max_depth=self.max_depth, **self._pipe_kw)
        self.pipe.inputs_resident = self.inputs_resident
        self.pipe.sdf_delta = self.sdf_delta
        return self.pipe

    # ---- phases ---------------------------------------------------------------------------------
    def encode(self, frame):
        """Encode stream: voxelise the whole frame (replicated), the exchange bounds to pinned memory, the point
        encoder on the pairs this rank owns, TSDF side fusion."""
        self.pointnet.shard = self.shard
        pipe = self._pipe_for(frame)
        return ShardFrame(slot=pipe.begin(frame), capacity=0, blocks=None, decode=False)

    def bound(self, fr):
        """Largest number of boundary records any rank can send for this frame (the same number on every rank)."""
        return self.pipe.bound(fr.slot) if self.world > 1 else 0

    def exchange_capacity(self, bound):
        """Records per rank the frame's all-gather moves: the bound rounded up to REC_QUANTUM, but never more than a
        slot's send block holds.  The bound counts TOUCHED boundary voxels (before the min-points filter) and the
        send block is sized for the voxels a frame can EMIT (8 * points / min_pts + 1), which is what a rank really
        sends: a small or sparse frame can have bound > send_cap, and the exchange must not run past the block.
        Every rank computes the same number (same bound, same frame sizes)."""
        if bound <= 0:
            return 0
        return min(-(-bound // REC_QUANTUM) * REC_QUANTUM, self.pipe.send_cap)

    def upsert(self, fr, capacity, decode=True):
        """Main stream: upsert of the owned voxels in ONE launch that also stamps them as the frame's decode origins
        and appends their boundary records to the slot's send block.  -> this rank's block (header + ``capacity``
        records, int32 words); None when nothing is exchanged."""
        fr.decode = decode
        send = self.pipe.upsert(fr.slot, decode=decode, ghost_rows=(self.world - 1) * capacity)
        if capacity == 0:
            return None
        if (capacity + 1) * REC_WORDS > send.numel():
            raise _lib.BnvError(f"exchange capacity {capacity} exceeds the slot's send block "
                                f"({send.numel() // REC_WORDS - 1} records)")
        return send[: (capacity + 1) * REC_WORDS]

    def recv_buffer(self, words):
        """The all-gather's output buffer (written and read on the main stream: one serves every frame)."""
        r = self._recv
        if r is None or r.numel() < words:
            r = self._recv = torch.empty(int(words * 1.5), dtype=torch.int32, device=self.dev)
        return r[:words]

    def install(self, fr, blocks, capacity):
        """blocks: [world * (capacity + 1) * REC_WORDS] int32 -- the all-gather's output; installed (and the send block
        reset) by the frame's finish call."""
        fr.blocks, fr.capacity = blocks, capacity
        return (self.world - 1) * capacity

    def decode(self, fr):
        return self.pipe.sdf[fr.slot]

    def finish(self, fr, sdf, reserved):
        self.pipe.finish(fr.slot, fr.blocks, fr.capacity)
        fr.blocks = None
        return fr

    def result(self, fr):
        """-> (coords [U'_r, 3] of the voxels this rank owns among the frame's, sdf [U'_r, 27]) or (None, None)."""
        w = self.pipe.result(fr.slot)
        from .pipeline import W_EVALS, W_COUNTERS
        self._last_evals = int(w[W_EVALS])
        self.last_owned_pairs = int(w[W_COUNTERS + 5])      # bnv_encode_counters_t.reserved[0]
        return self.pipe.outputs(fr.slot, w, copy=self.copy_results)

    def cancel(self, fr):
        """Abandons a frame whose encode was enqueued and that will not be upserted (bnv_frame_cancel)."""
        self.pipe.cancel(fr.slot)

    def owner_table(self):
        """The first-touch owner table (numpy uint8, one byte per block: bits 0..5 owner, bit 6 assigned, bit 7 touched)
        and the per-rank loads (uint64 [world]) as they stand now (host copies; synchronises), or (None, None)."""
        if self._owner_state is None:
            return None, None
        lib = self.volume._lib
        st = self._owner_state.cpu().numpy()
        n = np.asarray(self.volume._n_xyz_host, dtype=np.int64)
        nb = (n + (1 << self.block_log2) - 1) >> self.block_log2
        t0, l0 = int(lib.bnv_shard_state_table_offset()), int(lib.bnv_shard_state_loads_offset())
        return st[t0: t0 + int(nb.prod())].copy(), st[l0: l0 + 8 * self.world].view(np.uint64).copy()

    def owners(self, coords):
        """Host restatement of the ownership rule in force: coords [n, 3] -> rank [n]."""
        table, _ = self.owner_table()
        return voxel_owner(coords, self.world, self.block_log2, table, self.volume._n_xyz_host)

    def owned_rows_mask(self):
        """bool [rows]: rows this rank owns (the others are ghost rows)."""
        n = self.volume.num_rows()
        c = self.volume._row_coords[:n].cpu().numpy()
        return torch.from_numpy(self.owners(c) == self.rank).to(self.dev)

    def last_mlp_evals(self):
        return self.volume.last_lattice_evals()


class ShardHandle:
    """One frame of ShardedNeuralMap.  ``result()`` waits for that frame only; a failure is raised once and again on
    every later call (the frame's slot is free either way)."""

    def __init__(self, nm, fr):
        self._nm, self._fr, self._done, self._err = nm, fr, None, None

    @property
    def pending(self):
        return self._done is None and self._err is None

    def result(self):
        if self._err is not None:
            raise self._err
        if self._done is None:
            fr, self._fr = self._fr, None
            try:
                self._done = self._nm.backend.result(fr)
            except Exception as e:
                self._err = e
                raise
        return self._done


class ShardedNeuralMap:
    """Per-frame driver over one shard; every rank calls the same methods with the same frame.

    fuse_and_decode_async enqueues: encode (whole frame voxelised, owned voxels encoded; its own stream) -> upsert
    (+ boundary records) -> ONE all-gather -> install ghost rows -> lattice decode of the owned voxels.  One host
    wait (the exchange bound -- read from pinned memory; the encode stream reaches it while the main stream still
    works on the previous frame); the outputs are collected through the returned handle.  At most ``n_slots - 1``
    handles may stay uncollected (HIP backend; the oldest is collected on demand)."""

    def __init__(self, dimensions, voxel_size, pointnet, min_pts_in_grid=8, device="cuda:0", backend=None,
                 group=None, capacity=1 << 20, tsdf=False, ownership=None, block_log2=None, axis=None):
        import torch.distributed as dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = backend or HipShardBackend(dimensions, voxel_size, pointnet, self.rank, self.world,
                                                  min_pts_in_grid, capacity=capacity, device=device, tsdf=tsdf,
                                                  ownership=ownership, block_log2=block_log2, axis=axis)
        self.volume = getattr(self.backend, "volume", None)
        self.voxel_size = voxel_size
        self.exchanged_bytes = 0          # bytes this rank has received in all-gathers (statistics)
        self.host_waits = 0
        self._open = []                   # handles not collected yet, oldest first (HIP backend: they hold slots)
        self._pre = None                  # (frame, ShardFrame) whose encode was enqueued ahead (next_frame)

    def fuse_and_decode_async(self, frame, decode=True, next_frame=None):
        """``next_frame``: the frame the NEXT call will pass (the same object), if the caller knows it: its encode is
        enqueued now, BEFORE the host waits for this frame's exchange bound, so the encode stream always holds a frame
        more than the main stream and never idles while the host enqueues this frame's upsert .. decode."""
        import torch.distributed as dist
        be = self.backend
        ring = getattr(be, "n_slots", None)
        ahead = next_frame is not None and ring is not None
        if ring is not None:
            self._open = [h for h in self._open if h.pending]
            need = 1 + (1 if ahead else 0) - (1 if self._pre is not None else 0)
            while self._open and len(self._open) + (1 if self._pre is not None else 0) + need > ring:
                self._open.pop(0).result()            # the slot ring is full: collect the oldest frame
        if self._pre is not None and self._pre[0] is not frame:
            raise _lib.BnvError("fuse_and_decode_async: the frame announced as next_frame must be the next one passed "
                                "(flush() integrates an announced frame that will not come, abandon() drops it)")
        return self._frame(frame, decode, next_frame if ahead else None, ring)

    def _frame(self, frame, decode, next_frame, ring):
        be = self.backend
        fr = None
        upserted = False
        try:
            with torch.no_grad():
                if self._pre is not None:
                    fr, self._pre = self._pre[1], None
                else:
                    fr = be.encode(frame)
                if next_frame is not None:
                    self._pre = (next_frame, be.encode(next_frame))
                bound = be.bound(fr)                       # the frame's one host wait
                self.host_waits += 1
                capacity = (be.exchange_capacity(bound) if hasattr(be, "exchange_capacity")
                            else -(-bound // REC_QUANTUM) * REC_QUANTUM)
                send = be.upsert(fr, capacity, decode)
                upserted = True
                return self._exchange_and_finish(be, fr, send, capacity, decode, ring)
        except Exception:
            # Neither this frame (begun, not upserted: its slot would stay in state 1 and the ring would refuse it for
            # good) nor a frame announced and begun ahead may keep its slot when this one fails on the way.  A frame
            # that failed BEHIND its upsert cannot be taken back: its slot stays with the pipe (the volume holds it).
            if fr is not None and not upserted and hasattr(be, "cancel"):
                self._cancel(fr)
            self.abandon()
            raise

    def _cancel(self, fr):
        try:
            self.backend.cancel(fr)
        except Exception as e:      # (the original failure is what the caller must see; this one is reported)
            import warnings
            warnings.warn(f"ShardedNeuralMap: cancelling a begun frame failed as well: {e!r}")

    def _exchange_and_finish(self, be, fr, send, capacity, decode, ring):
        import torch.distributed as dist
        reserved = 0
        if capacity > 0:
            words = self.world * send.numel()
            recv = be.recv_buffer(words) if hasattr(be, "recv_buffer") else torch.empty(
                words, dtype=send.dtype, device=send.device)
            dist.all_gather_into_tensor(recv, send, group=self.group)      # THE collective of the frame
            self.exchanged_bytes += words * 4
            reserved = be.install(fr, recv, capacity)
        return self._finish(be, fr, decode, reserved, ring)

    def _finish(self, be, fr, decode, reserved, ring):
        sdf = be.decode(fr) if decode else None
        h = ShardHandle(self, be.finish(fr, sdf, reserved))
        if ring is not None:
            self._open.append(h)
        return h

    def fuse_and_decode(self, frame):
        return self.fuse_and_decode_async(frame).result()

    def integrate(self, frame):
        return self.fuse_and_decode_async(frame, decode=False).result()[0]

    def abandon(self):
        """Drops a frame that was announced as ``next_frame`` (its encode is enqueued) and will not be passed: nothing
        of it reaches the volume, its slot is free again.  Local, no collective -- but every rank must do the same
        (the ranks' volumes and owner tables stay equal only if all of them drop the frame)."""
        if self._pre is not None:
            fr, self._pre = self._pre[1], None
            if hasattr(self.backend, "cancel"):
                self._cancel(fr)

    def flush(self):
        """End of a stream: a frame announced as ``next_frame`` that was never passed is integrated now (no decode;
        COLLECTIVE like every frame), then every open handle is collected.  -> the results of the frames that were
        still open, oldest first."""
        last = None
        if self._pre is not None:
            last = self.fuse_and_decode_async(self._pre[0], decode=False)
        out = []
        if last is not None and last not in self._open:
            self._open.append(last)          # (backends without a slot ring keep no list: the flushed frame is collected too)
        while self._open:
            h = self._open.pop(0)
            if h.pending:
                out.append(h.result())
        return out

    close = flush

    def last_mlp_evals(self):
        """Device int32 [1]: SDF-MLP evaluations this rank ran for the last frame."""
        return self.backend.last_mlp_evals()


# =============================================================================================
# Frame-parallel mode: throughput scaling of one frame stream
# =============================================================================================
# One encoded frame travels in two pieces:
#   HEADER   8 int64 words = the 8 int32 device counters of bnv_encode_pointcloud in words 0..3
#            (n_valid, n_unique | n_out, n_avg bits | error, -); all-gathered first (64 B per rank), read by the host
#            while the GPU still works on the previous batch;
#   PAYLOAD  flat int64 [8 * rows], struct-of-arrays so that every section is a contiguous view the kernels read
#            in place:  [0, 3*rows) grid_ids [rows, 3] int64 | [3*rows, 4*rows) pcounts [rows] int64 |
#            [4*rows, 8*rows) feats [rows, 8] float32.  (One packing launch + ONE all-gather: measured faster than
#            all-gathering the encoder's three output arrays separately, every collective costs ~15 us.)
# rows = the largest n_out of the batch (from the headers, rounded up to ROW_QUANTUM), equal on every rank, so the
# payload all-gather moves what the batch really holds (64 B per emitted voxel) instead of the worst-case bound.
# Rows >= a frame's n_out are don't-care.
REC_HDR = 8
ROW_QUANTUM = 1024


def payload_words(rows):
    return 8 * int(rows)


def payload_views(p, rows):
    """-> (grid_ids [R, 3] i64, pcounts [R] i64, feats [R, 8] f32): views of a payload, no copies."""
    R = int(rows)
    return p[: 3 * R].view(R, 3), p[3 * R: 4 * R], p[4 * R: 8 * R].view(torch.float32).view(R, 8)


def header_counters(h):
    """int64 [8] header -> the 8 int32 encoder counters (view)."""
    return h[:4].view(torch.int32)


class EncodedFrame:
    """What encode_frame returns: the header and the encoder's capacity-sized output arrays."""

    def __init__(self, hdr, grid_ids=None, pcounts=None, feats=None):
        self.hdr, self.grid_ids, self.pcounts, self.feats = hdr, grid_ids, pcounts, feats


class HipFrameBackend:
    """Full (replicated) volume on one GPU; the per-frame phases as separate calls.  Nothing here waits for the
    GPU; the one host wait of a batch (its headers) is in FrameParallelNeuralMap.exchange."""

    def __init__(self, dimensions, voxel_size, pointnet, min_pts_in_grid=8, capacity=1 << 20, device="cuda:0",
                 tsdf=False, max_depth=3.0):
        from .sparse_volume import SparseVolume
        self.pointnet = pointnet
        self.volume = SparseVolume(8, voxel_size, dimensions, min_pts_in_grid, capacity=capacity, device=device)
        self.dev = self.volume._dev
        self.max_depth = max_depth          # the loader's depth cut-off (NeuralMap.max_depth)
        self.inputs_resident = False        # as NeuralMap.inputs_resident
        self.tsdf_vol = None
        if tsdf:                                               # run_e2e.py:60-71, as NeuralMap does
            from .sparse_volume import get_world_range
            from .tsdf import TSDFVolume
            mn, mx, _ = get_world_range(dimensions, 0.025)
            self.tsdf_vol = TSDFVolume(np.stack([mn, mx], 1), 0.025, device=device)
        self._scratch_ids = None
        # high priority: its few small launches are on the path of the next batch (measured: -35 us per batch)
        from .streams import concurrent_stream
        main = torch.cuda.current_stream(self.dev)
        self._side = concurrent_stream(self.dev, main, priority=-1)        # header / payload exchange: never behind the decode
        # the encode of batch k+1 depends on its frame only: on its own stream it runs beside batch k's upserts and
        # decode (its latency-bound kernels fill the decode's tail), as in NeuralMap.fuse_and_decode_async
        self.overlap_encode = True
        self._enc = concurrent_stream(self.dev, main, exclude=(self._side,))
        self._enc_src = None

    def record_rows(self, frame):
        """Upper bound of the voxels one frame can emit (every emitted voxel holds >= min_pts pairs)."""
        if "input_pts" in frame:
            n = int(frame["input_pts"].shape[1])
        else:
            n = int(frame["depth"].shape[-2] * frame["depth"].shape[-1])
        return 8 * n // max(self.pointnet.min_pts_in_grid, 1) + 1

    def encode_frame(self, frame):
        """Encodes one frame into capacity-sized arrays (the encoder's own outputs; nothing is copied)."""
        if self.overlap_encode and torch.cuda.current_stream() != self._enc:
            self._enc_src = self._enc
            if not self.inputs_resident:      # the frame's tensors may still be in production on the caller's stream
                self._enc.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._enc):
                return self.encode_frame(frame)
        if not self.overlap_encode:
            self._enc_src = None
        from .neural_map import frame_input_pts
        v = self.volume
        self.pointnet.shard = (0, 1, BLOCK_LOG2)
        rows = self.record_rows(frame)
        grid_ids = torch.empty((rows, 3), dtype=torch.int64, device=self.dev)
        pcounts = torch.empty(rows, dtype=torch.int64, device=self.dev)
        feats = torch.empty((rows, 8), dtype=torch.float32, device=self.dev)
        hdr = torch.empty(REC_HDR, dtype=torch.int64, device=self.dev)        # words 4..7 are never read
        if self._scratch_ids is None or self._scratch_ids.numel() < rows:
            self._scratch_ids = torch.empty(rows, dtype=torch.int64, device=self.dev)
        outs = (feats, pcounts, self._scratch_ids, grid_ids)
        if "input_pts" in frame:
            cnt = self.pointnet.encode_pointcloud_async(frame["input_pts"], v.n_xyz, v.min_coords, v.max_coords,
                                                        v.voxel_size, out=outs)[4]
        else:      # straight from the depth image: front end fused into the voxelisation
            cnt = self.pointnet.encode_depth_async(frame["depth"], frame["intr_mat"], frame["T_wc"], self.max_depth,
                                                   v.n_xyz, v.min_coords, v.max_coords, v.voxel_size, out=outs)[4]
        header_counters(hdr).copy_(cnt)
        return EncodedFrame(hdr, grid_ids, pcounts, feats)

    def empty_frame(self):
        return EncodedFrame(torch.zeros(REC_HDR, dtype=torch.int64, device=self.dev))

    def pack(self, enc, rows):
        """The first ``rows`` rows of an encoded frame as one payload (one launch)."""
        if enc.grid_ids is None:                 # this rank has no frame in the batch
            return torch.zeros(payload_words(rows), dtype=torch.int64, device=self.dev)
        for t in (enc.grid_ids, enc.pcounts, enc.feats):
            t.record_stream(torch.cuda.current_stream())
        have = int(enc.pcounts.shape[0])
        if rows <= have:
            return torch.cat([enc.grid_ids[:rows].reshape(-1), enc.pcounts[:rows],
                              enc.feats[:rows].reshape(-1).view(torch.int64)])
        p = torch.zeros(payload_words(rows), dtype=torch.int64, device=self.dev)   # another rank's frame is larger
        g, c, f = payload_views(p, rows)
        g[:have], c[:have], f[:have] = enc.grid_ids, enc.pcounts, enc.feats
        return p

    def side(self, after_main):
        """Context: the exchange stream.  ``after_main``: it first waits for what the main stream holds now."""
        if after_main:     # ... or the encode stream, when the encode was enqueued there
            self._side.wait_stream(self._enc_src if self._enc_src is not None else torch.cuda.current_stream())
        return torch.cuda.stream(self._side)

    def adopt(self, *tensors):
        """Tensors allocated on the exchange stream that main-stream kernels are about to read."""
        for t in tensors:
            if t is not None:
                t.record_stream(torch.cuda.current_stream())

    def integrate_record(self, hdr, payload, rows, n_out, frame=None):
        if n_out:
            grid_ids, pcounts, feats = payload_views(payload, rows)
            self.volume.integrate(grid_ids[:n_out], feats[:n_out], pcounts[:n_out], n_dev=header_counters(hdr)[2:3])
        if self.tsdf_vol is not None and frame is not None and "depth" in frame:
            self.tsdf_vol.integrate(frame.get("rgb"), frame["depth"], frame["intr_mat"], frame["T_wc"], obs_weight=1.,
                                    max_depth=self.max_depth)

    def integrate_records(self, hdr_all, out, rows, n_out, s0, s1):
        """_integrate of frames [s0, s1) of a gathered batch, in frame order, as ONE batched upsert (4 launches)."""
        items = []
        for s in range(s0, s1):
            if n_out[s]:
                grid_ids, pcounts, feats = payload_views(out[s], rows)
                k = n_out[s]
                items.append((grid_ids[:k], feats[:k], pcounts[:k], header_counters(hdr_all[s])[2:3]))
        if items:
            self.volume.integrate_batch(items)

    def integrate_tsdf(self, frames, n_valid=None):
        """TSDF side fusion of all frames of a batch (one launch per 8 frames; nothing reads it before the batch ends).
        ``n_valid``: the frames' in-bounds point counts (from the headers): frames without any are skipped, as the
        reference returns from NeuralMap.integrate before the TSDF fusion then (run_e2e.py:91-92)."""
        if self.tsdf_vol is None:
            return
        fr = [f for i, f in enumerate(frames) if "depth" in f and (n_valid is None or n_valid[i])]
        if fr:
            self.tsdf_vol.integrate_batch([f["depth"] for f in fr], [f["intr_mat"] for f in fr],
                                          [f["T_wc"] for f in fr], obs_weight=1., max_depth=self.max_depth,
                                          color_ims=[f.get("rgb") for f in fr])

    def decode_record(self, hdr, payload, rows, n_out):
        grid_ids, _, _ = payload_views(payload, rows)
        return self.volume.decode_lattice(grid_ids[:n_out], self.pointnet.nerf, None, query_tensor=False)

    def account(self, headers, n_rows_after=None):
        """Host bookkeeping of a batch: n_avg_pts statistics (sparse_volume.py:508-523) and the row reservations
        made for the capacity bound (the volume's row count behind the batch comes from a pinned read-back)."""
        for h in headers:
            c = header_counters(h)
            n_valid, n_out = int(c[0]), int(c[2])
            if n_out:
                self.volume.settle(n_out, n_rows_after if n_rows_after is not None else
                                   self.volume._rows_upper - self.volume._inflight)
            if n_valid:
                self.volume.track_n_pts(float(c[3:4].view(torch.float32)[0]))

    def pinned(self, shape):
        return torch.empty(shape, dtype=torch.int64, pin_memory=True)

    def rows_readback(self):
        """Pinned copy of {row count, sticky upsert error} behind everything enqueued so far."""
        return self.volume.status_readback()

    def event(self):
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def slice_result(self, payload, rows, sdf, n_out):
        grid_ids, _, _ = payload_views(payload, rows)
        return grid_ids[:n_out], None if sdf is None else sdf[:n_out]


class BatchHandle:
    """One batch of FrameParallelNeuralMap.  ``result()`` -> (coords [U', 3], sdf [U', 27]) of the frame
    THIS rank decoded, or (None, None); waits for that batch only."""

    def __init__(self, fp, payload, rows, sdf, host_hdr, event, n_frames, host_rows=None):
        self._fp, self._payload, self._rows, self._sdf = fp, payload, rows, sdf
        self._host, self._event, self._b, self._host_rows = host_hdr, event, n_frames, host_rows
        self._accounted = False
        self._done = None

    def _account(self):
        if not self._accounted:
            if self._event is not None:
                self._event.synchronize()
            n_after = int(self._host_rows[0]) if self._host_rows is not None else None
            self._fp.backend.account([self._host[s] for s in range(self._b)], n_after)
            self._accounted = True
            if self._host_rows is not None and len(self._host_rows) > 1 and int(self._host_rows[1]):
                self._fp.backend.volume.check_status(self._host_rows[1])   # device-side upsert failure: raise

    def result(self):
        if self._done is None:
            fp = self._fp
            while fp._unsettled and not self._accounted:      # bookkeeping stays in batch order
                fp._unsettled.pop(0)._account()
            self._account()
            c = header_counters(self._host[fp.rank])
            if self._payload is None or int(c[0]) == 0 or int(c[2]) == 0:
                self._done = (None, None)
            else:
                self._done = fp.backend.slice_result(self._payload, self._rows, self._sdf, int(c[2]))
            self._payload = self._sdf = None
        return self._done


class FrameParallelNeuralMap:
    """N ranks process a batch of up to N consecutive frames together:

      1. rank r encodes frame r of the batch (encode_pointcloud is a pure function of the frame);
      2. the 64-byte headers are all-gathered and copied to the host on a side stream; the host reads them while
         the GPU is still busy with the previous batch, sizes ONE all-gather of the payloads by the largest frame of
         the batch (64 B per emitted voxel) and starts it on the side stream;
      3. every rank replays _integrate (and the TSDF side fusion) for all frames of the batch IN FRAME
         ORDER on its replicated volume -- as batched upserts: frames 0..r, then the decode of frame r, then frames
         r+1.. (bnv_volume_integrate_batch: identical results to one upsert per frame).

    Every volume goes through exactly the single-GPU sequence of states, so all outputs equal the
    one-GPU run; decode (the largest kernel) and encode are spread over the ranks, only the cheap
    upserts are replicated.  ``process_stream`` software-pipelines the batches: batch k+1 is encoded and
    its exchange started (side stream + RCCL's own stream) BEFORE batch k is integrated and decoded, so the
    exchange overlaps the decode kernels."""

    def __init__(self, dimensions, voxel_size, pointnet, min_pts_in_grid=8, device="cuda:0", backend=None,
                 group=None, tsdf=False, capacity=1 << 20):
        import torch.distributed as dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = backend or HipFrameBackend(dimensions, voxel_size, pointnet, min_pts_in_grid, device=device,
                                                  tsdf=tsdf, capacity=capacity)
        self.volume = getattr(self.backend, "volume", None)
        self._last = None
        self.exchanged_bytes = 0   # payload bytes this rank has received in all-gathers (statistics)
        # The persistent MLP kernels fill every CU (1 workgroup each, most of the LDS and VGPRs), so an RCCL kernel
        # that becomes ready while one of them runs waits for its tail.  The exchange of batch k+1 is issued while
        # batch k's upserts run (small kernels, free CUs) and is needed a whole decode + encode later, and both MLP
        # kernels hand their tiles out dynamically, so workgroups displaced by the collective cost nothing but the
        # collective's own CU time.  Leaving CUs free permanently instead (BNV_OPTIONS="reserve_cus=n") costs 3-6 % of a
        # batch on one GPU (tools/fp_single_rank.py --reserve 8) and is therefore off by default.
        self._unsettled = []      # batches whose host bookkeeping has not been done yet (oldest first)
        self.max_unsettled = 3    # the host may run this many batches ahead of the GPU before it waits

    def submit(self, frames):
        """Encode this rank's frame of the batch and start the header exchange; returns a ticket."""
        import torch.distributed as dist
        b = len(frames)
        assert 1 <= b <= self.world
        be = self.backend
        with torch.no_grad():
            enc = be.encode_frame(frames[self.rank]) if self.rank < b else be.empty_frame()
            with be.side(after_main=True):
                hdr_all = torch.empty((self.world, REC_HDR), dtype=torch.int64, device=enc.hdr.device)
                dist.all_gather_into_tensor(hdr_all.view(-1), enc.hdr, group=self.group)
                host = be.pinned((self.world, REC_HDR))
                host.copy_(hdr_all, non_blocking=True)
                ev = be.event()
        return {"frames": frames, "enc": enc, "hdr": hdr_all, "host": host, "event": ev}

    def exchange(self, ticket):
        """Waits for the batch's headers (the only host wait of a batch), then starts the payload all-gather."""
        import torch.distributed as dist
        if "rows" in ticket:
            return ticket
        be = self.backend
        if ticket["event"] is not None:
            ticket["event"].synchronize()
        b = len(ticket["frames"])
        n_out, n_valid = [], []
        for s in range(b):
            c = header_counters(ticket["host"][s])
            if int(c[4]):
                from .fusion import encode_error_message
                raise _lib.BnvError(encode_error_message(int(c[4])))
            n_out.append(int(c[2]) if int(c[0]) else 0)
            n_valid.append(int(c[0]))
        rows = -(-max(n_out) // ROW_QUANTUM) * ROW_QUANTUM
        out = work = None
        if rows:
            with torch.no_grad(), be.side(after_main=False):
                send = be.pack(ticket["enc"], rows)
                out = torch.empty((self.world, payload_words(rows)), dtype=torch.int64, device=send.device)
                work = dist.all_gather_into_tensor(out.view(-1), send, group=self.group, async_op=True)
            self.exchanged_bytes += 8 * out.numel()
        ticket.update(rows=rows, n_out=n_out, n_valid=n_valid, out=out, work=work)
        return ticket

    def finish(self, ticket, decode=True):
        """Integrate the whole batch in frame order, decode this rank's frame; returns a BatchHandle."""
        be = self.backend
        self.exchange(ticket)
        frames, out, rows, n_out, hdr = ticket["frames"], ticket["out"], ticket["rows"], ticket["n_out"], ticket["hdr"]
        b = len(frames)
        if ticket["work"] is not None:
            ticket["work"].wait()
        be.adopt(out, hdr)
        # host bookkeeping of earlier batches, in order, without waiting: only batches whose event has already
        # fired (the host must be free to enqueue the N replayed integrates while the GPU still runs the encode)
        while self._unsettled and (self._unsettled[0]._event is None or self._unsettled[0]._event.query()
                                   or len(self._unsettled) >= self.max_unsettled):
            self._unsettled.pop(0)._account()
        with torch.no_grad():
            sdf = mine = None
            # frames up to and including this rank's own, then its decode (the state the single-GPU run decodes
            # from), then the rest of the batch: two batched upserts instead of one per frame
            own = min(self.rank + 1, b)
            be.integrate_records(hdr, out, rows, n_out, 0, own)
            if self.rank < b and n_out[self.rank]:
                mine = out[self.rank]
                if decode:
                    sdf = be.decode_record(hdr[self.rank], mine, rows, n_out[self.rank])
            be.integrate_records(hdr, out, rows, n_out, own, b)
            be.integrate_tsdf(frames, ticket["n_valid"])
            host_rows = be.rows_readback() if hasattr(be, "rows_readback") else None
            ev = be.event()
        self._last = BatchHandle(self, mine, rows, sdf, ticket["host"], ev, b, host_rows)
        self._unsettled.append(self._last)
        return self._last

    def process_batch(self, frames, decode=True):
        """frames: list of up to `world` frame dicts, the SAME list on every rank.
        Returns (coords, sdf) of the frame this rank decoded (None, None if it had none)."""
        return self.finish(self.submit(frames), decode).result()

    def process_stream(self, batches, decode=True):
        """batches: list of frame lists (each up to `world` long, the same on every rank).  Yields one
        BatchHandle per batch; batch k+1's encode is enqueued before batch k's integrate + decode, and its
        payload exchange is started right after them, as soon as its headers have reached the host."""
        ticket = self.exchange(self.submit(batches[0])) if batches else None
        for i in range(len(batches)):
            nxt = self.submit(batches[i + 1]) if i + 1 < len(batches) else None
            handle = self.finish(ticket, decode)
            if nxt is not None:
                self.exchange(nxt)
            yield handle
            ticket = nxt

    def flush(self):
        while self._unsettled:
            self._unsettled.pop(0)._account()

    def _dev(self):
        return getattr(self.backend, "dev", torch.device("cpu"))
