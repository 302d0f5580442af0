"""SparseVolume -- drop-in for the reference class of the same name
(src/models/sparse_volume.py:484-892), backed by the HIP hash volume (csrc/volume.hip) and the
HIP decode kernels (csrc/decode.hip).  Same constructor, attributes and method names; tensors
live on the GPU; torch only owns the memory.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def get_world_range(dimensions, voxel_size):
    """voxel_utils.py:83-88 (float64 numpy, one voxel of padding each side)."""
    dimensions = np.asarray(dimensions, dtype=np.float64)
    min_ = -dimensions / 2 - voxel_size
    max_ = dimensions / 2 + voxel_size
    n_xyz = np.ceil((max_ - min_) / voxel_size).astype(int).tolist()
    max_ = min_ + voxel_size * np.asarray(n_xyz)
    return min_, max_, n_xyz


def make_grid(n_xyz, bound_min, bound_max, voxel_size, min_pts_in_grid, shard=(0, 1, 3), mlp_mode=None):
    """bnv_grid_t with the float32 values the reference compares against: ``bound_max - voxel_size``
    and ``bound_min + voxel_size`` are float32-tensor (op) python-float results
    (local_point_fusion.py:94-100), evaluated here with the same torch CPU ops."""
    bmin = torch.as_tensor(bound_min).detach().float().cpu().reshape(3)
    bmax = torch.as_tensor(bound_max).detach().float().cpu().reshape(3)
    lo = bmin + voxel_size
    hi = bmax - voxel_size
    g = _lib.Grid()
    for a in range(3):
        g.bound_min[a] = float(bmin[a])
        g.bound_lo[a] = float(lo[a])
        g.bound_hi[a] = float(hi[a])
        g.n_xyz[a] = int(n_xyz[a])
    g.voxel_size = float(np.float32(voxel_size))
    g.min_pts_in_grid = int(min_pts_in_grid)
    g.shard_rank, g.shard_world, g.shard_block_log2 = int(shard[0]), int(shard[1]), int(shard[2])
    g.shard_state = int(shard[3]) if len(shard) > 3 and shard[3] else None     # first-touch owner table (device pointer)
    g.mlp_mode = 0 if mlp_mode is None else int(mlp_mode) + 1      # 0: the library's process default
    return g


def _pow2_at_least(x):
    p = 1
    while p < x:
        p *= 2
    return p


class SparseVolume:
    # Dense row index of the grid ("brick", include/bnv_fusion.h): kept automatically for grids up to this many
    # voxels (4 B each: 64 MB at 256^3, 512 MB at 512^3); ``brick=True / False`` in the constructor overrides.  The
    # hash is always complete, so a volume without the index only decodes a little slower.
    BRICK_MAX_VOXELS = 1 << 27

    def __init__(self, n_feats, voxel_size, dimensions, min_pts_in_grid, capacity=100000, device="cuda:0",
                 brick=None):
        min_coords, max_coords, n_xyz = get_world_range(dimensions, voxel_size)
        self.device = device
        self._dev = torch.device(device)
        self._lib = _lib.require_device(self._dev.index or 0)
        self.dimensions = dimensions
        self.voxel_size = voxel_size
        self.min_coords = torch.from_numpy(min_coords).float().to(device)   # sparse_volume.py:495
        self.max_coords = torch.from_numpy(max_coords).float().to(device)
        self.n_xyz = torch.from_numpy(np.asarray(n_xyz)).long().to(device)  # sparse_volume.py:497
        self._n_xyz_host = [int(v) for v in n_xyz]
        self.n_feats = n_feats
        if n_feats != 8:
            raise ValueError("the HIP volume stores 8-float features (model.feature_vector_size=8)")
        self.min_pts_in_grid = min_pts_in_grid
        self.shard = (0, 1, 3)
        self._grid = make_grid(n_xyz, min_coords, max_coords, voxel_size, min_pts_in_grid, self.shard)
        self._grid_modes = {}      # copies of _grid per arithmetic mode (_grid_for)
        self._want_brick = brick
        self._brick = None
        self._ws = None
        self._slot_mask = None        # side tables of integrate_batch (per slot; re-made with the slot table)
        self._slot_items = None
        self._lattice_ws = None
        self._lattice_ws2 = None      # second decode workspace (frame pipeline: alternating frames)
        self._lattice_last = None     # the one handed out last
        self._stamp = None
        self._stamped = None          # (epoch, origins pointer, n) of an integrate(..., stamp_origins=True)
        self._epoch = 0
        self._lws_generation = 0      # counts re-makes of the decode workspaces (the frame pipeline forgets old pointers)
        # persistent lattice tables of the frame pipeline (include/bnv_fusion.h: bnv_volume_t.lattice_table): made on
        # demand (enable_persistent_tables), re-made zeroed with the row arrays.  Every method of this class that writes
        # features clears the rows' have-words on the device; what the entries ALSO depend on -- the SDF network's
        # weights and arithmetic mode -- is watched by the frame pipe (FramePipe.finish: mode, model, pack version)
        self._ptable = None
        self._phave = None
        self.reset(capacity)
        self.avg_n_pts = 0
        self.n_pts_list = []
        self.n_frames = 0
        self.min_pts = 1000
        self.max_pts = 0

    # ---- bookkeeping (sparse_volume.py:508-523) ------------------------------------------------
    def track_n_pts(self, n_pts):
        n_pts = float(n_pts)
        self.n_pts_list.append(n_pts)
        self.avg_n_pts = (self.avg_n_pts * self.n_frames + n_pts) / (self.n_frames + 1)
        self.n_frames += 1
        self.min_pts = min(self.min_pts, n_pts)
        self.max_pts = max(self.max_pts, n_pts)

    def print_statistic(self):
        print("===========")
        p = np.percentile(self.n_pts_list, [25, 50, 75]) if self.n_pts_list else [0, 0, 0]
        self.per_25, self.per_50, self.per_75 = p[0], p[1], p[2]
        print(f"25%: {p[0]}, 50%: {p[1]}, 75%:{p[2]}")
        print(f"mean: {self.avg_n_pts}, min: {self.min_pts}, max:{self.max_pts}")
        print("===========")

    # ---- storage ---------------------------------------------------------------------------------
    def reset(self, capacity):
        """sparse_volume.py:587-600."""
        cap = max(int(capacity), 1024)
        d = self._dev
        self._row_capacity = cap
        self._n_slots = _pow2_at_least(2 * cap)
        self._slot_keys = torch.empty(self._n_slots, dtype=torch.int64, device=d)
        self._slot_rows = torch.empty(self._n_slots, dtype=torch.int32, device=d)
        self._row_coords = torch.zeros((cap, 3), dtype=torch.int64, device=d)
        self._features = torch.zeros((cap, 8), dtype=torch.float32, device=d)
        self._weights = torch.zeros(cap, dtype=torch.float32, device=d)
        self._num_hits = torch.zeros(cap, dtype=torch.float32, device=d)
        # dense row index of the grid (include/bnv_fusion.h: bnv_volume_t.brick): 4 B per voxel of the grid, filled
        # with -1 by bnv_volume_clear below; the allocation survives reset(); no memory for it -> go without
        nvox = self._n_xyz_host[0] * self._n_xyz_host[1] * self._n_xyz_host[2]
        want = (nvox <= self.BRICK_MAX_VOXELS) if self._want_brick is None else bool(self._want_brick)
        if not want or nvox >= (1 << 31):
            self._brick = None
        elif self._brick is None or self._brick.numel() != nvox:
            try:
                self._brick = torch.empty(nvox, dtype=torch.int32, device=d)
            except torch.OutOfMemoryError:
                self._brick = None
        self._status = torch.zeros(2, dtype=torch.int32, device=d)   # {rows in use, sticky upsert error}
        self._n_rows = self._status[:1]
        self._rows_upper = 0          # host-side upper bound of *n_rows (avoids a sync per insert)
        self._inflight = 0            # rows reserved by enqueued, not yet settled, device-count integrates
        self._rows_known = 0          # largest row count read back so far
        self._lattice_ws = None
        self._lattice_ws2 = None
        self._lattice_last = None
        self._stamp = None
        self._stamped = None          # (epoch, origins pointer, n) of an integrate(..., stamp_origins=True)
        if self._phave is not None:
            self._make_persistent_tables()
        _lib.check(self._lib.bnv_volume_clear(C.byref(self._struct()), _lib.stream_ptr()), "bnv_volume_clear")
        self.tensor_indexer = None
        self.features = None
        self.weights = None
        self.num_hits = None
        self.active_coordinates = None
        self._snapshot_rows = 0

    def _struct(self):
        v = _lib.Volume()
        v.slot_keys = self._slot_keys.data_ptr()
        v.slot_rows = self._slot_rows.data_ptr()
        v.n_slots = self._n_slots
        v.row_coords = self._row_coords.data_ptr()
        v.features = self._features.data_ptr()
        v.weights = self._weights.data_ptr()
        v.num_hits = self._num_hits.data_ptr()
        v.row_capacity = self._row_capacity
        v.n_rows = self._n_rows.data_ptr()
        v.n_feats = 8
        v.brick = self._brick.data_ptr() if self._brick is not None else None
        if self._phave is not None:      # (every kernel that writes features clears the row's word; lattice_persist
            v.lattice_table = self._ptable.data_ptr()      # stays 0: only the frame pipeline decodes from the tables)
            v.lattice_have = self._phave.data_ptr()
        for a in range(3):
            v.brick_dims[a] = self._n_xyz_host[a]
        return v

    def num_rows(self):
        """Exact number of active voxels (one device->host read)."""
        st = self._status.tolist()            # the stream is drained up to here: nothing is in flight any more
        n = int(st[0])
        self._rows_known = max(self._rows_known, n)
        self._rows_upper = n + self._inflight
        self.check_status(st[1])
        return n

    UPSERT_ERRORS = {1: "hash table full", 2: "voxel coordinate outside the 21-bit key range",
                     3: "row capacity exceeded", 4: "a rank's boundary-record block overflowed (sharded exchange)",
                     5: "feature values beyond the range certified for the f16-split SDF decoder (or NaN): "
                        "decode in exact fp32 -- bnv_fusion_amd.set_mlp_mode(0)"}

    def check_status(self, err):
        """Raises on the sticky error word of the upsert kernels (device int32 next to the row counter; the
        asynchronous pipelines read it back with the row count, see status_readback)."""
        err = int(err)
        if err:
            raise _lib.BnvError("volume status: " + self.UPSERT_ERRORS.get(err, f"error {err}"))

    def status_readback(self):
        """Pinned int32 [2] = {row count, sticky upsert error} behind everything enqueued so far (async copy)."""
        h = torch.empty(2, dtype=torch.int32, pin_memory=True)
        h.copy_(self._status, non_blocking=True)
        return h

    def settle(self, n_reserved, n_rows_after):
        """Bookkeeping of an integrate that was enqueued with a device-side count: ``n_reserved`` rows had been
        set aside for it; ``n_rows_after`` is the volume's row count read back (pinned copy) behind it.  Keeps
        the host-side bound exact without ever synchronising: bound = last known count + what is in flight."""
        self._inflight -= int(n_reserved)
        self._rows_known = max(self._rows_known, int(n_rows_after))     # row counts only grow
        self._rows_upper = self._rows_known + self._inflight

    def release(self, n_reserved):
        """An enqueued integrate whose result was never collected: its reservation is no longer "in flight", but
        how many rows it created is unknown, so the host-side bound keeps the whole reservation until the next exact
        read (num_rows()) -- the bound must stay an upper bound."""
        self._inflight -= int(n_reserved)

    def _reserve(self, n_new):
        """Grow rows / slot table so that n_new more keys fit (the Open3D map auto-grows).  The test uses the
        host-side BOUND (known rows + reservations of integrates still in flight), and growth provides for the
        bound too -- otherwise every call would have to synchronise to learn that the real count still fits."""
        if self._rows_upper + n_new <= self._row_capacity:
            return
        n = self.num_rows()
        need = n + self._inflight + n_new
        if need <= self._row_capacity:
            return
        cap = max(2 * self._row_capacity, need)
        d = self._dev
        # the arrays and workspaces are re-made: streams other than the current one (the frame pipeline's blend /
        # encode streams) may still be reading the old ones
        torch.cuda.synchronize(d)

        def grow(t, shape):
            o = torch.zeros(shape, dtype=t.dtype, device=d)
            o[:n] = t[:n]
            return o

        self._row_coords = grow(self._row_coords, (cap, 3))
        self._features = grow(self._features, (cap, 8))
        self._weights = grow(self._weights, (cap,))
        self._num_hits = grow(self._num_hits, (cap,))
        self._row_capacity = cap
        self._n_slots = _pow2_at_least(2 * cap)
        self._slot_keys = torch.empty(self._n_slots, dtype=torch.int64, device=d)
        self._slot_rows = torch.empty(self._n_slots, dtype=torch.int32, device=d)
        self._lattice_ws = None
        self._lattice_ws2 = None
        self._lattice_last = None
        self._stamp = None
        self._stamped = None          # (epoch, origins pointer, n) of an integrate(..., stamp_origins=True)
        if self._phave is not None:
            self._make_persistent_tables()
        _lib.check(self._lib.bnv_volume_rehash(C.byref(self._struct()), _lib.stream_ptr()), "bnv_volume_rehash")

    def _make_persistent_tables(self):
        self._ptable = torch.empty(self._row_capacity * 27, dtype=torch.float32, device=self._dev)
        self._phave = torch.zeros(self._row_capacity, dtype=torch.int32, device=self._dev)

    def enable_persistent_tables(self):
        """The frame pipeline's persistent lattice tables (112 bytes per row of capacity): SDF table entries of rows a
        frame did not update are carried over from the frame that computed them."""
        if self._phave is None:
            self._make_persistent_tables()

    def invalidate_tables(self):
        """Forgets every persistent table entry (on the current stream).  Needed after writing the volume's feature
        ROWS by any means other than this class's methods -- ``to_tensor()`` returns a copy, and the optimiser's way back
        (``insert``) clears the rows' words itself -- and after changing the SDF network the entries were computed with
        (the frame pipe does that itself: it watches the model object, its arithmetic mode and its pack version)."""
        if self._phave is not None:
            self._phave.zero_()

    def _workspace(self, n):
        need = int(self._lib.bnv_volume_workspace_bytes(int(n)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(max(need, 1 << 20), dtype=torch.uint8, device=self._dev)
        return self._ws

    # ---- reference API ---------------------------------------------------------------------------
    def integrate(self, coords, feats, pcounts, n_dev=None, stamp_origins=False):
        """Fused LitFusionPointNet._integrate (local_point_fusion.py:647-673): query + running
        average + upsert for UNIQUE keys in one pass.  ``n_dev`` (device int32 [1]): take the element
        count from the device; the tensors are then capacity-sized buffers (no host sync needed).
        ``stamp_origins``: the same launch marks the rows it touches as the origins of the NEXT lattice decode, which
        must be ``decode_lattice(coords, ..., prestamped=True)`` on the same stream (one launch less per frame)."""
        n = int(coords.shape[0])
        self._stamped = None
        if n == 0:
            return
        coords = coords.reshape(-1, 3).long().contiguous()
        feats = feats.float().contiguous()
        pcounts = pcounts.reshape(-1).long().contiguous()
        self._reserve(n)
        ws = self._workspace(n)
        if stamp_origins:
            lws, epoch = self._lattice_workspace(n)          # (behind _reserve: growth re-makes this workspace)
            ex = _lib.IntegrateExtras()
            ex.lattice_ws = lws.data_ptr()
            ex.stamp_epoch = epoch
            _lib.check(self._lib.bnv_volume_integrate_frame(C.byref(self._struct()), _lib.ptr(coords), _lib.ptr(feats),
                                                            _lib.ptr(pcounts), n, _lib.ptr(n_dev), _lib.ptr(ws),
                                                            ws.numel(), C.byref(ex), _lib.stream_ptr()),
                       "bnv_volume_integrate_frame")
            self._stamped = (epoch, coords.data_ptr(), n)
        else:
            _lib.check(self._lib.bnv_volume_integrate(C.byref(self._struct()), _lib.ptr(coords), _lib.ptr(feats),
                                                      _lib.ptr(pcounts), n, _lib.ptr(n_dev), _lib.ptr(ws), ws.numel(),
                                                      _lib.stream_ptr()), "bnv_volume_integrate")
        self._rows_upper += n
        if n_dev is not None:
            self._inflight += n

    BATCH_MAX = 8     # BNV_VOLUME_BATCH_MAX

    def integrate_batch(self, frames):
        """``integrate`` for several consecutive frames at once: ``frames`` is a list of (coords, feats, pcounts,
        n_dev) tuples in frame order (n_dev as in ``integrate``, may be None).  Two launches per BATCH_MAX frames;
        rows, features and weights come out exactly as from one ``integrate`` call per frame."""
        items = []
        for coords, feats, pcounts, n_dev in frames:
            n = int(coords.shape[0])
            if n:
                items.append((coords.reshape(-1, 3).long().contiguous(), feats.float().contiguous(),
                              pcounts.reshape(-1).long().contiguous(), n_dev, n))
        for g0 in range(0, len(items), self.BATCH_MAX):
            grp = items[g0: g0 + self.BATCH_MAX]
            total = sum(it[4] for it in grp)
            self._reserve(total)                       # may re-make the slot table: side tables after it
            if self._slot_mask is None or self._slot_mask.numel() != self._n_slots:
                self._slot_mask = torch.zeros(self._n_slots, dtype=torch.int32, device=self._dev)
                self._slot_items = torch.empty(self._n_slots * self.BATCH_MAX, dtype=torch.int32, device=self._dev)
            ws = self._workspace(sum((it[4] + 255) // 256 * 256 for it in grp))
            k = len(grp)
            arr = lambda j: (C.c_void_p * k)(*[it[j].data_ptr() for it in grp])
            nd = (C.c_void_p * k)(*[(it[3].data_ptr() if it[3] is not None else 0) for it in grp])
            ns = (C.c_int64 * k)(*[it[4] for it in grp])
            _lib.check(self._lib.bnv_volume_integrate_batch(
                C.byref(self._struct()), k, arr(0), arr(1), arr(2), ns, nd, _lib.ptr(self._slot_mask),
                _lib.ptr(self._slot_items), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "bnv_volume_integrate_batch")
            self._rows_upper += total
            self._inflight += sum(it[4] for it in grp if it[3] is not None)

    def insert(self, keys, new_feats, new_weights, new_num_hits):
        """sparse_volume.py:561-585 (upsert)."""
        if len(keys) == 0:
            return None
        keys = keys.reshape(-1, 3).long().contiguous()
        n = int(keys.shape[0])
        f = new_feats.detach().float().reshape(n, 8).contiguous()
        w = new_weights.detach().float().reshape(n).contiguous()
        h = new_num_hits.detach().float().reshape(n).contiguous()
        self._reserve(n)
        ws = self._workspace(n)
        _lib.check(self._lib.bnv_volume_insert(C.byref(self._struct()), _lib.ptr(keys), _lib.ptr(f), _lib.ptr(w),
                                               _lib.ptr(h), n, _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                   "bnv_volume_insert")
        self._rows_upper += n

    def to_tensor(self):
        """sparse_volume.py:525-559: snapshot (copy) of the active entries, in buffer order."""
        n = self.num_rows()
        self.active_coordinates = self._row_coords[:n].clone()
        self.features = self._features[:n].clone()
        self.weights = self._weights[:n].clone().unsqueeze(-1)
        self.num_hits = self._num_hits[:n].clone().unsqueeze(-1)
        self._snapshot_rows = n
        self.tensor_indexer = True
        return self.active_coordinates, self.features, self.weights, self.num_hits

    def _lookup(self, keys, feats, weights, hits, limit):
        shapes = list(keys.shape)
        assert shapes[-1] == 3
        n = int(np.prod(shapes[:-1]))
        k = keys.reshape(-1, 3).long().contiguous()
        of = torch.empty((n, 8), dtype=torch.float32, device=self._dev)
        ow = torch.empty((n, 1), dtype=torch.float32, device=self._dev)
        oh = torch.empty((n, 1), dtype=torch.float32, device=self._dev)
        _lib.check(self._lib.bnv_volume_query(C.byref(self._struct()), _lib.ptr(k), n, _lib.ptr(feats),
                                              _lib.ptr(weights), _lib.ptr(hits), int(limit), _lib.ptr(of),
                                              _lib.ptr(ow), _lib.ptr(oh), None, _lib.stream_ptr()),
                   "bnv_volume_query")
        return (of.reshape(shapes[:-1] + [8]), ow.reshape(shapes[:-1] + [1]), oh.reshape(shapes[:-1] + [1]))

    def query(self, keys):
        """sparse_volume.py:661-695: live values, zeros for absent keys."""
        if int(np.prod(list(keys.shape)[:-1])) == 0:
            return None, None, None
        return self._lookup(keys, self._features, self._weights, self._num_hits, self._row_capacity)

    def _snapshot(self):
        assert self.features is not None, "call self.to_tensor() first."
        f = self.features.detach()
        w = self.weights.detach()
        h = self.num_hits.detach()
        if not (f.is_contiguous() and w.is_contiguous() and h.is_contiguous()):
            raise ValueError("volume.features / weights / num_hits must stay contiguous")
        return f, w, h, min(int(f.shape[0]), self._snapshot_rows)

    def _query_tensor(self, keys):
        """sparse_volume.py:625-659: values of the to_tensor() snapshot."""
        f, w, h, lim = self._snapshot()
        return self._lookup(keys, f, w, h, lim)

    def count_optim(self, keys):
        """sparse_volume.py:602-622."""
        f, w, h, lim = self._snapshot()
        k = keys.reshape(-1, 3).long().contiguous()
        if self._stamp is None or self._stamp.numel() < self._row_capacity:
            self._stamp = torch.zeros(self._row_capacity, dtype=torch.int32, device=self._dev)
        self._epoch += 1
        _lib.check(self._lib.bnv_volume_count_optim(C.byref(self._struct()), _lib.ptr(k), int(k.shape[0]),
                                                    _lib.ptr(w), lim, _lib.ptr(self._stamp), self._epoch,
                                                    _lib.stream_ptr()), "bnv_volume_count_optim")

    def count_optim_pts(self, pts, is_coords=False):
        """count_optim on the 8 corner voxels of sample points [..., 3] (world units unless is_coords) -- what
        render_utils.py:488-493 does via get_neighbors, without materialising the int64 keys."""
        f, w, h, lim = self._snapshot()
        p = pts.detach().reshape(-1, 3).float().contiguous()
        if self._stamp is None or self._stamp.numel() < self._row_capacity:
            self._stamp = torch.zeros(self._row_capacity, dtype=torch.int32, device=self._dev)
        self._epoch += 1
        _lib.check(self._lib.bnv_volume_count_optim_pts(C.byref(self._struct()), C.byref(self._grid), _lib.ptr(p),
                                                        int(p.shape[0]), 1 if is_coords else 0, _lib.ptr(w), lim,
                                                        _lib.ptr(self._stamp), self._epoch, _lib.stream_ptr()),
                   "bnv_volume_count_optim_pts")

    # ---- all ray splits of an optimiser step at once (include/bnv_fusion.h: bnv_optim_step) -------------------------
    def _split_mask(self):
        """uint32 per row of the to_tensor() snapshot, zero between steps (bnv_volume_apply_split_counts clears it)."""
        m = getattr(self, "_split_mask_buf", None)
        if m is None or m.numel() < self._row_capacity:
            m = self._split_mask_buf = torch.zeros(self._row_capacity, dtype=torch.int32, device=self._dev)
        return m

    def count_optim_splits(self, pts, split_samples, is_coords=False):
        """count_optim (sparse_volume.py:602-622) of every split of a step, deferred: records in the split mask which
        splits touch which row (sample q of ``pts`` belongs to split q // split_samples); the weights change when
        ``apply_split_counts`` is called, behind the decodes that need the per-split values."""
        f, w, h, lim = self._snapshot()
        p = pts.detach().reshape(-1, 3).float().contiguous()
        _lib.check(self._lib.bnv_volume_count_optim_splits(
            C.byref(self._struct()), C.byref(self._grid), _lib.ptr(p), int(p.shape[0]), 1 if is_coords else 0, lim,
            int(split_samples), _lib.ptr(self._split_mask()), _lib.stream_ptr()), "bnv_volume_count_optim_splits")

    def apply_split_counts(self):
        f, w, h, lim = self._snapshot()
        _lib.check(self._lib.bnv_volume_apply_split_counts(_lib.ptr(w), _lib.ptr(self._split_mask()), lim,
                                                           _lib.stream_ptr()), "bnv_volume_apply_split_counts")

    def optim_step(self, pts, nerf, sdf_delta, split_samples, target, sample_weight, n_valid, loss2, grad_features,
                   pred=None):
        """Forward + L1 ray loss + backward of all samples of a step in ONE launch (fp32 decoders).  ``loss2``: float
        [2], zero on entry ([0] accumulates the loss, [1] is the kernel's work counter)."""
        grid = self._grid_for(nerf)
        f, w, _, lim = self._snapshot()
        c = pts.detach().reshape(-1, 3).float().contiguous()
        d, keep = self._delta(sdf_delta)
        assert grad_features.shape == f.shape and grad_features.is_contiguous()
        _lib.check(self._lib.bnv_optim_step(
            C.byref(self._struct()), C.byref(grid), _lib.ptr(f), _lib.ptr(w), int(lim), _lib.ptr(nerf.sdf_pack),
            _lib.ptr(nerf.sdf_bwd_pack), _lib.ptr(c), int(c.shape[0]), 0, C.byref(d), _lib.ptr(self._split_mask()),
            int(split_samples), _lib.ptr(target), _lib.ptr(sample_weight), _lib.ptr(n_valid), _lib.ptr(loss2),
            _lib.ptr(pred), _lib.ptr(grad_features), _lib.stream_ptr()), "bnv_optim_step")

    def decode_pts_splits(self, pts, nerf, sdf_delta, split_samples):
        """bnv_decode_pts whose mask decisions see the deferred count_optim of the splits (tiny-cuda-nn decoders, tests)."""
        grid = self._grid_for(nerf)
        c = pts.detach().reshape(-1, 3).float().contiguous()
        f, w, _, lim = self._snapshot()
        d, keep = self._delta(sdf_delta)
        out = torch.empty(int(c.shape[0]), dtype=torch.float32, device=self._dev)
        _lib.check(self._lib.bnv_decode_pts_splits(
            C.byref(self._struct()), C.byref(grid), _lib.ptr(f), _lib.ptr(w), int(lim), _lib.ptr(nerf.sdf_pack),
            _lib.ptr(c), int(c.shape[0]), 0, C.byref(d), _lib.ptr(self._split_mask()), int(split_samples),
            _lib.ptr(out), _lib.stream_ptr()), "bnv_decode_pts_splits")
        return out

    def decode_pts_backward_splits(self, pts, nerf, grad_sdf, grad_features, split_samples):
        grid = self._grid_for(nerf)
        f, w, _, lim = self._snapshot()
        c = pts.detach().reshape(-1, 3).float().contiguous()
        g = grad_sdf.detach().reshape(-1).float().contiguous()
        assert grad_features.shape == f.shape and grad_features.is_contiguous()
        _lib.check(self._lib.bnv_decode_pts_backward_splits(
            C.byref(self._struct()), C.byref(grid), _lib.ptr(f), _lib.ptr(w), int(lim), _lib.ptr(nerf.sdf_pack),
            _lib.ptr(nerf.sdf_bwd_pack), _lib.ptr(c), int(c.shape[0]), 0, _lib.ptr(self._split_mask()),
            int(split_samples), _lib.ptr(g), _lib.ptr(grad_features), _lib.stream_ptr()),
            "bnv_decode_pts_backward_splits")

    def decode_pts_backward(self, coords, nerf, grad_sdf, grad_features, is_coords=False):
        """Accumulates d(sum(grad_sdf * decode_pts(coords))) / d features into ``grad_features`` [M, 8] (the
        to_tensor() snapshot rows).  The autograd edge of decode_pts calls this; the fused optimiser step too."""
        if not hasattr(nerf, "sdf_bwd_pack"):
            raise NotImplementedError("decode_pts backward needs a decoder with sdf_bwd_pack")
        grid = self._grid_for(nerf)
        f, w, _, lim = self._snapshot()
        c = coords.detach().reshape(-1, 3).float().contiguous()
        g = grad_sdf.detach().reshape(-1).float().contiguous()
        assert grad_features.shape == f.shape and grad_features.is_contiguous()
        _lib.check(self._lib.bnv_decode_pts_backward(
            C.byref(self._struct()), C.byref(grid), _lib.ptr(f), _lib.ptr(w), int(lim),
            _lib.ptr(nerf.sdf_pack), _lib.ptr(nerf.sdf_bwd_pack), _lib.ptr(c), int(c.shape[0]),
            1 if is_coords else 0, _lib.ptr(g), _lib.ptr(grad_features), _lib.stream_ptr()), "bnv_decode_pts_backward")

    # ---- decode ------------------------------------------------------------------------------------
    def _delta(self, sdf_delta):
        d = _lib.SdfDelta()
        keep = None
        if sdf_delta is not None:
            keep = sdf_delta.detach().float().contiguous()
            assert keep.dim() == 5 and keep.shape[0] == 1 and keep.shape[1] == 1
            d.data = keep.data_ptr()
            for a in range(3):
                d.dims[a] = int(keep.shape[2 + a])
        return d, keep

    def _grid_for(self, nerf):
        """The volume's bnv_grid_t for calls that run ``nerf``'s networks: the arithmetic mode travels in the grid."""
        return _lib.grid_with_mode(self._grid, _lib.model_mode(nerf), self._grid_modes)

    def _values(self, query_tensor):
        if query_tensor:
            f, w, _, lim = self._snapshot()
            return f, w, lim
        return self._features, self._weights, self._row_capacity

    def decode_pts(self, coords, nerf, sdf_delta=None, is_coords=False, query_tensor=True):
        """sparse_volume.py:768-833.  coords [1, B, S, 3] -> [1, B, S, 1].

        When ``volume.features`` requires grad (run_e2e.py:114 wraps it in nn.Parameter for the global
        optimiser) and ``query_tensor`` is set, the result is differentiable w.r.t. the features: the
        backward is the HIP kernel behind ``bnv_decode_pts_backward``."""
        if (query_tensor and torch.is_grad_enabled() and self.features is not None
                and self.features.requires_grad):
            shape = list(coords.shape)
            out = _DecodePts.apply(self.features, self, coords, nerf, sdf_delta, bool(is_coords))
            return out.reshape(shape[:-1] + [1])
        return self._decode_pts_forward(coords, nerf, sdf_delta, is_coords, query_tensor)

    def _decode_pts_forward(self, coords, nerf, sdf_delta, is_coords, query_tensor):
        grid = self._grid_for(nerf)
        shape = list(coords.shape)
        c = coords.detach().reshape(-1, 3).float().contiguous()
        n = int(c.shape[0])
        f, w, lim = self._values(query_tensor)
        d, keep = self._delta(sdf_delta)
        out = torch.empty(n, dtype=torch.float32, device=self._dev)
        _lib.check(self._lib.bnv_decode_pts(C.byref(self._struct()), C.byref(grid), _lib.ptr(f), _lib.ptr(w),
                                            int(lim), _lib.ptr(nerf.sdf_pack), _lib.ptr(c), n,
                                            1 if is_coords else 0, C.byref(d), _lib.ptr(out), _lib.stream_ptr()),
                   "bnv_decode_pts")
        return out.reshape(shape[:-1] + [1])

    def decode_lattice(self, origins, nerf, sdf_delta=None, query_tensor=True, n_dev=None, prestamped=False):
        """decode_pts on the 3x3x3 lattice {-0.5, 0, 0.5}^3 around integer voxel ``origins`` [B, 3]
        (the decode SparseVolume.meshlize performs, sparse_volume.py:717-738) -> [B, 27].
        ``prestamped``: ``origins`` is the very tensor the last ``integrate(..., stamp_origins=True)`` upserted."""
        grid = self._grid_for(nerf)
        o = origins.detach().reshape(-1, 3).long().contiguous()
        n = int(o.shape[0])
        out = torch.empty((n, 27), dtype=torch.float32, device=self._dev)
        if n == 0:
            return out
        f, w, lim = self._values(query_tensor)
        d, keep = self._delta(sdf_delta)
        stamped = getattr(self, "_stamped", None)
        self._stamped = None
        if prestamped:
            if stamped != (self._lattice_epoch, o.data_ptr(), n) or self._lattice_ws is None:
                raise _lib.BnvError("decode_lattice(prestamped=True) needs the origins of the integrate(..., "
                                    "stamp_origins=True) right before it")
            _lib.check(self._lib.bnv_decode_lattice_stamped(
                C.byref(self._struct()), C.byref(grid), _lib.ptr(f), _lib.ptr(w), int(lim),
                _lib.ptr(nerf.sdf_pack), _lib.ptr(o), n, _lib.ptr(n_dev), C.byref(d), _lib.ptr(self._lattice_ws),
                self._lattice_ws.numel(), self._lattice_epoch, _lib.ptr(out), _lib.stream_ptr()),
                "bnv_decode_lattice_stamped")
            return out
        self._lattice_workspace(n)
        _lib.check(self._lib.bnv_decode_lattice(C.byref(self._struct()), C.byref(grid), _lib.ptr(f),
                                                _lib.ptr(w), int(lim), _lib.ptr(nerf.sdf_pack), _lib.ptr(o), n,
                                                _lib.ptr(n_dev), C.byref(d), _lib.ptr(self._lattice_ws),
                                                self._lattice_ws.numel(),
                                                self._lattice_epoch, _lib.ptr(out), _lib.stream_ptr()),
                   "bnv_decode_lattice")
        return out

    def _lattice_workspace(self, n, which=0):
        """(workspace of the lattice decode for up to n voxels, a fresh epoch).  Re-made (zero-filled) when the volume
        grows: its front part is indexed by row.  ``which`` = 1: a second workspace -- the frame pipeline alternates
        two, so that a frame's blend (on a stream of its own) and the next frame's marking never share one."""
        need = int(self._lib.bnv_decode_lattice_workspace_bytes(int(n), self._row_capacity))
        if which:
            # (which = 1, 2, ..: further workspaces)
            if self._lattice_ws2 is None:
                self._lattice_ws2 = {}
            ws = self._lattice_ws2.get(which)
            if ws is None or ws.numel() < need:
                ws = self._lattice_ws2[which] = torch.zeros(int(need * 1.25) + 4096, dtype=torch.uint8, device=self._dev)
                self._lws_generation += 1
            self._lattice_epoch += 1           # (one counter for all: each workspace sees increasing epochs)
            self._lattice_last = ws
            return ws, self._lattice_epoch
        if self._lattice_ws is None or self._lattice_ws.numel() < need:
            # zero-filled: the per-row stamps at the front of the workspace must start at 0
            self._lattice_ws = torch.zeros(int(need * 1.25) + 4096, dtype=torch.uint8, device=self._dev)
            self._lws_generation += 1
            if self._lattice_ws2 is None:
                self._lattice_epoch = 0
        self._lattice_epoch += 1
        self._lattice_last = self._lattice_ws
        return self._lattice_ws, self._lattice_epoch

    def last_lattice_table_rows(self):
        """Device int32 tensor [1]: rows listed by the last bnv_lattice_neighbors(build_list) (sharded
        path: 27 MLP evaluations each)."""
        off = int(self._lib.bnv_decode_lattice_count_offset(self._row_capacity))
        return self._lattice_last[off: off + 4].view(torch.int32)

    def last_lattice_evals(self):
        """Device int32 tensor [1]: SDF-MLP evaluations of the last decode_lattice call (table entries
        read by live lattice points)."""
        off = int(self._lib.bnv_decode_lattice_count_offset(self._row_capacity))
        return self._lattice_last[off + 4: off + 8].view(torch.int32)

    def meshlize(self, nerf, sdf_delta=None, path=None):
        """sparse_volume.py:697-766: decode the 3x3x3 lattice of every active voxel and run per-voxel
        marching cubes -- both on the GPU.  Returns (active_pts, mesh) like the reference (None when no
        voxel straddles the surface); ``mesh`` is a bnv_fusion_amd.mesh.TriMesh (vertices / faces /
        export), standing in for trimesh.Trimesh(process=False)."""
        from .mesh import TriMesh, marching_cubes_lattice_indexed, to_host
        assert self.active_coordinates is not None, "call self.to_tensor() first."
        active_pts = self.active_coordinates * self.voxel_size + self.min_coords
        sdf = self.decode_lattice(self.active_coordinates, nerf, sdf_delta, query_tensor=True)
        # per voxel (verts, faces) with shared vertices, concatenated as the reference does (:740-756)
        verts, faces, _, _ = marching_cubes_lattice_indexed(sdf, self.active_coordinates, self.voxel_size,
                                                            self.min_coords)
        if faces.shape[0] == 0:
            return None
        v_host, f_host, pts_host = to_host(verts, faces, active_pts)
        mesh = TriMesh(v_host, f_host)
        if path is not None:
            mesh.export(path)
        return pts_host, mesh

    def meshlize_sdf(self, nerf, sdf_delta=None):
        """The decode half of meshlize only: (active_pts, sdf [M, 3, 3, 3]) on the device."""
        assert self.active_coordinates is not None, "call self.to_tensor() first."
        active_pts = self.active_coordinates * self.voxel_size + self.min_coords
        sdf = self.decode_lattice(self.active_coordinates, nerf, sdf_delta, query_tensor=True)
        return active_pts, sdf.reshape(-1, 3, 3, 3)

    def save(self, path):
        self.print_statistic()
        n = self._snapshot_rows

        def stat(v):
            # the reference's statistics are 0-d float32 device tensors once a frame has been tracked (track_n_pts
            # is handed the encode's n_avg_pts tensor, sparse_volume.py:508-513); plain numbers before that
            return torch.tensor(float(v), dtype=torch.float32, device=self._dev) if self.n_frames else v

        out_dict = {
            "25%": getattr(self, "per_25", None), "50%": getattr(self, "per_50", None),
            "75%": getattr(self, "per_75", None), "dimensions": self.dimensions,
            "voxel_size": self.voxel_size, "mean": stat(self.avg_n_pts), "min": stat(self.min_pts),
            "active_keys": self.active_coordinates, "active_vals": torch.arange(n, device=self._dev)[:, None],
            "features": self.features, "weights": self.weights, "num_hits": self.num_hits,
            "active_coordinates": self.active_coordinates,
        }
        torch.save(out_dict, path + "_sparse_volume.pth")

    def load(self, path):
        # the reference's file format (a dict with numpy entries) predates torch's weights_only default
        volume = torch.load(path, map_location=self._dev, weights_only=False)
        coords = volume["active_coordinates"].to(self._dev)
        self.reset(max(len(coords), 1024))
        self.insert(coords, volume["features"].to(self._dev), volume["weights"].to(self._dev),
                    volume["num_hits"].to(self._dev))
        self.to_tensor()


class _DecodePts(torch.autograd.Function):
    """SparseVolume.decode_pts as an autograd node whose only differentiable input is the
    ``to_tensor()`` feature table (what render_utils.py:493-497 back-propagates into)."""

    @staticmethod
    def forward(ctx, features, volume, coords, nerf, sdf_delta, is_coords):
        if not hasattr(nerf, "sdf_bwd_pack"):
            raise NotImplementedError("decode_pts backward is implemented for the fp32 decoder (LocalNeRFModel)")
        out = volume._decode_pts_forward(coords, nerf, sdf_delta, is_coords, True)
        ctx.volume, ctx.nerf, ctx.is_coords = volume, nerf, is_coords
        ctx.save_for_backward(features, coords.detach().reshape(-1, 3).float().contiguous(),
                              volume.weights.detach().clone())
        return out.reshape(-1)

    @staticmethod
    def backward(ctx, grad_out):
        features, c, w = ctx.saved_tensors
        volume, nerf = ctx.volume, ctx.nerf
        grid = volume._grid_for(nerf)
        f = features.detach()
        lim = min(int(f.shape[0]), volume._snapshot_rows)
        g = grad_out.detach().reshape(-1).float().contiguous()
        grad_f = torch.zeros_like(f)
        _lib.check(volume._lib.bnv_decode_pts_backward(
            C.byref(volume._struct()), C.byref(grid), _lib.ptr(f), _lib.ptr(w), int(lim),
            _lib.ptr(nerf.sdf_pack), _lib.ptr(nerf.sdf_bwd_pack), _lib.ptr(c), int(c.shape[0]),
            1 if ctx.is_coords else 0, _lib.ptr(g), _lib.ptr(grad_f), _lib.stream_ptr()), "bnv_decode_pts_backward")
        return grad_f, None, None, None, None, None


class VolumeList:
    """Drop-in for the reference's VolumeList (sparse_volume.py:895-1158): a thin wrapper around one
    ``fine_volume`` used by the Lightning test path (local_point_fusion.py:773-799, 801-864)."""

    def __init__(self, n_feats, voxel_size, dimensions, min_pts_in_grid, capacity=100000, device="cuda:0"):
        self.fine_volume = SparseVolume(n_feats, voxel_size, dimensions, min_pts_in_grid, capacity, device)
        self.fine_min_coords = self.fine_volume.min_coords
        self.fine_max_coords = self.fine_volume.max_coords
        self.fine_n_xyz = self.fine_volume.n_xyz
        self.fine_voxel_size = voxel_size
        self.device = device

    def to_tensor(self):
        (self.fine_active_coords, self.fine_feats, self.fine_weights,
         self.fine_num_hits) = self.fine_volume.to_tensor()

    def query(self, keys):
        return self.fine_volume.query(keys)

    def insert(self, keys, new_feats, new_weights, new_num_hits):
        self.fine_volume.insert(keys, new_feats, new_weights, new_num_hits)

    def decode_pts(self, pts, nerf, sdf_delta=None, query_tensor=True):
        """sparse_volume.py:1123-1150: world points -> SDF.  The world -> voxel conversion
        ``(pts - min) / voxel`` happens inside the kernel with the reference's two fp32 roundings."""
        return self.fine_volume.decode_pts(pts, nerf, sdf_delta=sdf_delta[0] if sdf_delta is not None else None,
                                           is_coords=False, query_tensor=query_tensor)

    def meshlize_coords(self, coords, nerf, sdf_delta=None, volume_resolution=None):
        """sparse_volume.py:970-1032 up to (not including) marching cubes: the SDF lattice [n, 3, 3, 3] of
        the given voxel coordinates that exist in the volume, decoded from live values.  The reference
        converts the lattice to world coordinates and back in fp32 (:1001, :1141), which perturbs the
        exact half-integer lattice by an ulp; that path is reproduced by decode_pts(pts) -- here the
        exact lattice is used (SparseVolume.meshlize's formulation)."""
        c = coords.reshape(-1, 3).long()
        _, w, _ = self.fine_volume.query(c)
        present = w[:, 0] > 0
        c = c[present]
        sdf = self.fine_volume.decode_lattice(c, nerf, sdf_delta[0] if sdf_delta is not None else None,
                                              query_tensor=False)
        return c, sdf.reshape(-1, 3, 3, 3)

    def meshlize(self, nerf, sdf_delta=None, volume_resolution=None, path=None):
        """sparse_volume.py:1034-1121 -> trimesh-like mesh (the reference drops active_pts here)."""
        out = self.fine_volume.meshlize(nerf, sdf_delta[0] if sdf_delta is not None else None, path)
        return None if out is None else out[1]

    def save(self, path):
        self.fine_volume.save(path + "_fine")

    def load(self, path):
        self.fine_volume.load(path)
