"""Multi-GPU local fusion + decode: the active-voxel set sharded by spatial hash, one process per
GPU, RCCL over xGMI (SURVEY.md section 8e -- new design; the reference is single-GPU).

Ownership   owner(voxel) = mix64(block coordinate) % world, blocks of 8^3 voxels (same function in
            csrc/bnv_common.hpp: voxel_owner).  A (point, corner) pair contributes to exactly one
            voxel, so every rank voxelises the whole frame (cheap, replicated) but runs the point
            encoder only on pairs whose voxel it owns: per-voxel sums are complete locally, NO
            reduction collective, results identical to one GPU.
Exchange    decode of a touched voxel reads the 27-entry SDF tables and weights of its 3x3x3
            neighbour voxels, some owned elsewhere.  Per frame: (1) all-gather of the touched voxel
            coordinates (each rank contributes the ones it owns), (2) every rank evaluates the SDF
            MLP for the rows IT owns among the neighbours of all touched voxels, (3) one all-gather
            of those records (coords 24 B + weight 4 B + table 108 B), installed on every rank as
            halo rows, (4) each rank blends the lattice of the touched voxels it owns.  Outputs
            stay sharded.  Both all-gathers are variable-size (sizes first, then padded payload).

The frame logic lives in ``ShardedNeuralMap`` and talks to a backend; ``HipShardBackend`` is the
product backend (HIP kernels).  The phases are exposed separately so that tests can drive several
shards in one process.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

BLOCK_LOG2 = 3


def mix64(k):
    k = np.asarray(k, dtype=np.uint64).copy()
    with np.errstate(over="ignore"):
        k ^= k >> np.uint64(33)
        k *= np.uint64(0xff51afd7ed558ccd)
        k ^= k >> np.uint64(33)
        k *= np.uint64(0xc4ceb9fe1a85ec53)
        k ^= k >> np.uint64(33)
    return (k & np.uint64(0xFFFFFFFF)).astype(np.uint64)


def voxel_owner(coords, world, block_log2=BLOCK_LOG2):
    """Host restatement of csrc/bnv_common.hpp voxel_owner: coords [n, 3] int -> rank [n]."""
    c = np.asarray(coords, dtype=np.int64)
    if world <= 1:
        return np.zeros(len(c), dtype=np.int64)
    b = (c >> block_log2).astype(np.uint64) & np.uint64(0xFFFFFFFF)
    key = (b[:, 0] << np.uint64(42)) | (b[:, 1] << np.uint64(21)) | b[:, 2]
    return (mix64(key) % np.uint64(world)).astype(np.int64)


def all_gather_var(t, group=None):
    """All-gather of tensors whose first dimension differs per rank -> concatenation in rank order."""
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


class HipShardBackend:
    """One shard of the volume on one GPU, HIP kernels."""

    def __init__(self, dimensions, voxel_size, pointnet, rank, world, min_pts_in_grid=8, capacity=1 << 20,
                 device="cuda:0"):
        from .sparse_volume import SparseVolume
        self.pointnet = pointnet
        self.rank, self.world = rank, world
        pointnet.shard = (rank, world, BLOCK_LOG2)
        self.volume = SparseVolume(8, voxel_size, dimensions, min_pts_in_grid, capacity=capacity, device=device)
        self.dev = self.volume._dev
        self._ghost = torch.zeros(self.volume._row_capacity, dtype=torch.uint8, device=self.dev)
        self._epoch = 0

    # ---- helpers --------------------------------------------------------------------------------
    def _ws(self, n):
        v = self.volume
        need = int(v._lib.bnv_decode_lattice_workspace_bytes(max(int(n), 1), v._row_capacity))
        if v._lattice_ws is None or v._lattice_ws.numel() < need:
            v._lattice_ws = torch.zeros(int(need * 1.25) + 4096, dtype=torch.uint8, device=self.dev)
            self._epoch = 0
        if self._ghost.numel() < v._row_capacity:
            g = torch.zeros(v._row_capacity, dtype=torch.uint8, device=self.dev)
            g[: self._ghost.numel()] = self._ghost
            self._ghost = g
        return v._lattice_ws

    def _table(self):
        v = self.volume
        off = int(v._lib.bnv_decode_lattice_table_offset(v._row_capacity))
        return v._lattice_ws[off: off + v._row_capacity * 27 * 4].view(torch.float32).view(v._row_capacity, 27)

    def rows_of(self, keys):
        v = self.volume
        k = keys.reshape(-1, 3).long().contiguous()
        rows = torch.empty(k.shape[0], dtype=torch.int32, device=self.dev)
        _lib.check(v._lib.bnv_volume_query(C.byref(v._struct()), _lib.ptr(k), int(k.shape[0]), _lib.ptr(v._features),
                                           _lib.ptr(v._weights), _lib.ptr(v._num_hits), v._row_capacity, None, None,
                                           None, _lib.ptr(rows), _lib.stream_ptr()), "bnv_volume_query")
        return rows.long()

    # ---- phases ---------------------------------------------------------------------------------
    def encode_integrate(self, frame):
        v = self.volume
        self.pointnet.shard = (self.rank, self.world, BLOCK_LOG2)
        from .neural_map import frame_input_pts
        f, c, _, coords, n_avg = self.pointnet.encode_pointcloud(
            frame_input_pts(frame), v.n_xyz, v.min_coords, v.max_coords, v.voxel_size, return_dense=False)
        if f is None:
            return torch.zeros((0, 3), dtype=torch.int64, device=self.dev)
        v.track_n_pts(n_avg)
        v.integrate(coords, f, c)
        return coords

    def tables_for(self, touched):
        """Records (coords [m,3] i64, weights [m], table [m,27]) of the rows this shard owns among the
        neighbours of ``touched`` (the global touched set)."""
        v = self.volume
        n = int(touched.shape[0])
        ws = self._ws(n)
        self._epoch += 1
        t = touched.reshape(-1, 3).long().contiguous()
        _lib.check(v._lib.bnv_lattice_neighbors(C.byref(v._struct()), C.byref(v._grid), _lib.ptr(v._weights),
                                                v._row_capacity, _lib.ptr(t), n, None, _lib.ptr(self._ghost), 1,
                                                _lib.ptr(ws), ws.numel(), self._epoch, _lib.stream_ptr()),
                   "bnv_lattice_neighbors")
        _lib.check(v._lib.bnv_lattice_table(C.byref(v._struct()), C.byref(v._grid), _lib.ptr(v._features),
                                            _lib.ptr(self.pointnet.nerf.sdf_pack), n, 0, _lib.ptr(ws), ws.numel(),
                                            _lib.stream_ptr()), "bnv_lattice_table")
        m = int(v.last_lattice_table_rows().item())
        off = int(v._lib.bnv_decode_lattice_list_offset(max(n, 1), v._row_capacity))
        rows = ws[off: off + 4 * m].view(torch.int32).long()
        return v._row_coords[rows], v._weights[rows], self._table()[rows]

    def install_and_blend(self, owned_touched, rec_coords, rec_weights, rec_tables):
        """Installs all exchanged records (halo rows for foreign ones) and blends the lattice of the
        touched voxels this shard owns -> [n, 27]."""
        v = self.volume
        n = int(owned_touched.shape[0])
        out = torch.empty((n, 27), dtype=torch.float32, device=self.dev)
        mine = torch.from_numpy(voxel_owner(rec_coords.cpu().numpy(), self.world) == self.rank).to(self.dev)
        foreign = ~mine
        if bool(foreign.any()):
            fk = rec_coords[foreign]
            z = torch.zeros((fk.shape[0], 8), dtype=torch.float32, device=self.dev)
            v.insert(fk, z, rec_weights[foreign], torch.zeros(fk.shape[0], device=self.dev))
        ws = self._ws(max(n, 1))
        rows = self.rows_of(rec_coords)
        self._table()[rows] = rec_tables
        self._ghost[rows[foreign]] = 1
        if n == 0:
            return out
        self._epoch += 1
        o = owned_touched.reshape(-1, 3).long().contiguous()
        _lib.check(v._lib.bnv_lattice_neighbors(C.byref(v._struct()), C.byref(v._grid), _lib.ptr(v._weights),
                                                v._row_capacity, _lib.ptr(o), n, None, None, 0, _lib.ptr(ws), ws.numel(),
                                                self._epoch, _lib.stream_ptr()), "bnv_lattice_neighbors")
        d = _lib.SdfDelta()
        _lib.check(v._lib.bnv_lattice_blend(C.byref(v._struct()), C.byref(v._grid), _lib.ptr(o), n, None, C.byref(d),
                                            _lib.ptr(ws), ws.numel(), _lib.ptr(out), _lib.stream_ptr()),
                   "bnv_lattice_blend")
        return out

    def owned_rows_mask(self):
        n = self.volume.num_rows()
        return self._ghost[:n] == 0


class ShardedNeuralMap:
    """Per-frame driver over one shard; every rank calls the same methods with the same frame."""

    def __init__(self, dimensions, voxel_size, pointnet, min_pts_in_grid=8, device="cuda:0", backend=None,
                 group=None):
        import torch.distributed as dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = backend or HipShardBackend(dimensions, voxel_size, pointnet, self.rank, self.world,
                                                  min_pts_in_grid, device=device)
        self.volume = getattr(self.backend, "volume", None)
        self.voxel_size = voxel_size

    def integrate(self, frame):
        with torch.no_grad():
            return self.backend.encode_integrate(frame)

    def fuse_and_decode(self, frame):
        with torch.no_grad():
            owned = self.backend.encode_integrate(frame)
            touched = all_gather_var(owned, self.group)                      # collective 1: coordinates
            rc, rw, rt = self.backend.tables_for(touched)
            rec = all_gather_var(torch.cat([rw.reshape(-1, 1), rt], dim=1), self.group)   # collective 2: tables
            rec_c = all_gather_var(rc, self.group)
            sdf = self.backend.install_and_blend(owned, rec_c, rec[:, 0].contiguous(), rec[:, 1:].contiguous())
        return owned, sdf


# =============================================================================================
# Frame-parallel mode: throughput scaling of one frame stream
# =============================================================================================
class HipFrameBackend:
    """Full (replicated) volume on one GPU; the three per-frame phases as separate calls."""

    def __init__(self, dimensions, voxel_size, pointnet, min_pts_in_grid=8, capacity=1 << 20, device="cuda:0"):
        from .sparse_volume import SparseVolume
        self.pointnet = pointnet
        self.volume = SparseVolume(8, voxel_size, dimensions, min_pts_in_grid, capacity=capacity, device=device)
        self.dev = self.volume._dev

    def encode(self, frame):
        """-> (coords [n,3] i64, counts [n] i64, feats [n,8] f32, n_avg float tensor) of one frame."""
        v = self.volume
        self.pointnet.shard = (0, 1, BLOCK_LOG2)
        from .neural_map import frame_input_pts
        f, c, _, g, n_avg = self.pointnet.encode_pointcloud(frame_input_pts(frame), v.n_xyz, v.min_coords,
                                                            v.max_coords, v.voxel_size, return_dense=False)
        if f is None:
            z = torch.zeros
            return (z((0, 3), dtype=torch.int64, device=self.dev), z(0, dtype=torch.int64, device=self.dev),
                    z((0, 8), device=self.dev), z((), device=self.dev))
        return g, c.reshape(-1), f, n_avg

    def integrate(self, coords, counts, feats, n_avg):
        if coords.shape[0] == 0:
            return
        self.volume.track_n_pts(n_avg)
        self.volume.integrate(coords, feats, counts)

    def decode(self, coords):
        return self.volume.decode_lattice(coords, self.pointnet.nerf, None, query_tensor=False)


class FrameParallelNeuralMap:
    """N ranks process a batch of up to N consecutive frames together:

      1. rank r encodes frame r of the batch (encode_pointcloud is a pure function of the frame);
      2. ONE variable-size all-gather of the encoded voxels (coords, count, 8 features = 64 B each,
         plus one header row per rank carrying n_avg_pts);
      3. every rank replays _integrate for all frames of the batch IN FRAME ORDER on its replicated
         volume, and decodes the lattice of frame r right after integrating frame r.

    Every volume goes through exactly the single-GPU sequence of states, so all outputs equal the
    one-GPU run; decode (the largest kernel) and encode are spread over the ranks, only the cheap
    upserts are replicated."""

    def __init__(self, dimensions, voxel_size, pointnet, min_pts_in_grid=8, device="cuda:0", backend=None,
                 group=None):
        import torch.distributed as dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = backend or HipFrameBackend(dimensions, voxel_size, pointnet, min_pts_in_grid, device=device)
        self.volume = getattr(self.backend, "volume", None)

    @staticmethod
    def _pack(coords, counts, feats, n_avg):
        n = coords.shape[0]
        rec = torch.zeros((n + 1, 8), dtype=torch.int64, device=coords.device)
        rec[0, 0] = n
        rec[0, 1:2] = n_avg.reshape(1).float().view(torch.int32).long()
        rec[1:, :3] = coords
        rec[1:, 3] = counts
        rec[1:, 4:] = feats.contiguous().view(torch.int64).reshape(n, 4)
        return rec

    @staticmethod
    def _unpack(rec):
        coords = rec[1:, :3].contiguous()
        counts = rec[1:, 3].contiguous()
        feats = rec[1:, 4:].contiguous().view(torch.float32).reshape(-1, 8)
        n_avg = rec[0, 1:2].int().view(torch.float32)[0]
        return coords, counts, feats, n_avg

    def process_batch(self, frames, decode=True):
        """frames: list of up to `world` frame dicts, the SAME list on every rank.
        Returns (coords, sdf) of the frame this rank decoded (None, None if it had none)."""
        import torch.distributed as dist
        b = len(frames)
        assert 1 <= b <= self.world
        with torch.no_grad():
            if self.rank < b:
                rec = self._pack(*self.backend.encode(frames[self.rank]))
            else:
                rec = torch.zeros((0, 8), dtype=torch.int64, device=self._dev())
            sizes = torch.zeros(self.world, dtype=torch.int64, device=rec.device)
            sizes[self.rank] = rec.shape[0]
            dist.all_reduce(sizes, group=self.group)
            sizes = sizes.tolist()
            m = max(max(sizes), 1)
            pad = torch.zeros((m, 8), dtype=torch.int64, device=rec.device)
            pad[: rec.shape[0]] = rec
            out = torch.empty((self.world, m, 8), dtype=torch.int64, device=rec.device)
            dist.all_gather_into_tensor(out.view(-1, 8), pad, group=self.group)
            mine = (None, None)
            for s in range(b):
                coords, counts, feats, n_avg = self._unpack(out[s, : sizes[s]])
                self.backend.integrate(coords, counts, feats, n_avg)
                if decode and s == self.rank:
                    mine = (coords, self.backend.decode(coords))
        return mine

    def _dev(self):
        return getattr(self.backend, "dev", torch.device("cpu"))
