"""FramePipe -- the per-frame chain (NeuralMap.integrate, run_e2e.py:78-109, + the lattice decode of the frame's voxels,
sparse_volume.py:697-738) behind the C object of csrc/pipeline.hip: persistent per-slot buffers, two HIP streams, no
per-frame allocation, event object or size read on the host.  Used by the spatially sharded map
(distributed.HipShardBackend) and usable on one GPU (world 1).

A frame occupies a slot from ``begin`` to ``result``:

    slot = pipe.begin(frame)              # encode stream: front end, voxelise, rank, [bound -> pinned], PointNet, TSDF
    bound = pipe.bound(slot)              # host wait (0 when unsharded)
    send = pipe.upsert(slot, decode, ghost_rows)   # main stream: upsert (+ boundary records, + origin stamps)
    ... all-gather of send[: (capacity + 1) * REC_WORDS] on the main stream ...
    pipe.finish(slot, blocks, capacity)   # main stream: install, lattice decode, read-backs
    words = pipe.result(slot)             # host wait; then pipe.outputs(slot, words)
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

REC_WORDS = 12          # BNV_SHARD_RECORD_BYTES / 4
HOST_WORDS = 96         # BNV_PIPE_HOST_WORDS
W_COUNTERS, W_STATUS, W_EVALS, W_BOUNDS = 0, 8, 10, 16


class FramePipe:
    # CUs the persistent point encoder leaves to the small kernels of the other streams in the four-stream schedule
    # (csrc/pipeline.hip): a share of 1 / 4 on a sharded volume, where a rank's encoder launch is small and the main
    # stream's chain (upsert -> exchange -> install -> mark) runs beside it; all CUs otherwise.  The constructor's
    # ``encoder_workgroups`` / BNV_PIPE_ENCODER_WGS override (0 = all CUs).
    ENCODER_SHARE_SHARDED = 0.75

    def __init__(self, volume, pointnet, max_points, n_slots=4, tsdf_vol=None, max_depth=3.0, sdf_delta=None,
                 streams=4, encoder_workgroups=None, persistent_tables=None):
        from .frontend import DEPTH_DTYPES
        self._dtypes = DEPTH_DTYPES
        self.volume, self.pointnet, self.tsdf_vol = volume, pointnet, tsdf_vol
        self.max_depth = float(max_depth)
        self.sdf_delta = sdf_delta
        v = volume
        dev = v._dev
        self.dev = dev
        self._lib = v._lib
        lib = self._lib
        self.n_slots = int(n_slots)
        assert 1 <= self.n_slots <= 8
        self.max_points = int(max_points)
        self.world = int(v.shard[1])
        from .streams import pipe_streams
        import os
        self.main = torch.cuda.current_stream(dev)
        streams = int(streams)
        # side streams verified to overlap the main stream and one another, shared by the pipes of the process
        side = pipe_streams(dev, self.main, 3 if streams >= 4 else 1)
        self.enc = side[0]
        # the front end (voxelise + rank) and the blend on streams of their own (csrc/pipeline.hip: why four);
        # streams=2: round 3's two-stream schedule (kept for the A/B in tests and tools).  The schedules measured
        # slower -- table MLP on a fifth stream from a feature snapshot, CU-masked streams, an early exchange of
        # contribution records, a gated encoder, a high-priority main stream -- were removed in round 6; their records
        # are profiles/r04_cu_mask_experiment.txt, r04_fifth_stream_experiment.txt and r05_experiments.txt [e8], [e9].
        self.front = self.blend = None
        if streams >= 4:
            self.front, self.blend = side[1], side[2]
        self.double_buffered = self.front is not None
        if encoder_workgroups is None:
            encoder_workgroups = os.environ.get("BNV_PIPE_ENCODER_WGS")
            if encoder_workgroups is None:
                cus = int(lib.bnv_num_compute_units())
                encoder_workgroups = int(cus * self.ENCODER_SHARE_SHARDED) if (self.world > 1 and streams >= 4) else 0
        self.encoder_workgroups = int(encoder_workgroups)
        res = v._n_xyz_host
        nvox = res[0] * res[1] * res[2]
        self.cap = max(min(8 * self.max_points // max(pointnet.min_pts_in_grid, 1) + 1, nvox), 1)
        self.send_cap = self.cap if self.world > 1 else 0
        n_arr = (C.c_int32 * 3)(*res)
        need = int(lib.bnv_encode_workspace_bytes(self.max_points, n_arr))
        self._enc_ws = torch.zeros(need, dtype=torch.uint8, device=dev)          # zero-filled = clean
        self._enc_ws2 = torch.zeros(need, dtype=torch.uint8, device=dev) if self.double_buffered else None
        cap, S = self.cap, self.n_slots
        self.input_pts = torch.empty((S, self.max_points, 6), dtype=torch.float32, device=dev)
        self.feats = torch.empty((S, cap, 8), dtype=torch.float32, device=dev)
        self.pcounts = torch.empty((S, cap), dtype=torch.int64, device=dev)
        self.flat_ids = torch.empty((S, cap), dtype=torch.int64, device=dev)
        self.grid_ids = torch.empty((S, cap, 3), dtype=torch.int64, device=dev)
        self.counters = torch.zeros((S, 8), dtype=torch.int32, device=dev)
        self.sdf = torch.empty((S, cap, 27), dtype=torch.float32, device=dev)
        self.send = None
        if self.world > 1:
            self.send = torch.zeros((S, (self.send_cap + 1) * REC_WORDS), dtype=torch.int32, device=dev)
            self.send[:, 1] = int(v.shard[0])                                     # header {count 0, sender rank, 0}
        self.host = torch.zeros((S, HOST_WORDS), dtype=torch.int32).pin_memory()
        self._host_np = self.host.numpy()
        cfg = _lib.FramePipeConfig()
        self._mode = _lib.model_mode(pointnet)
        self._grid = _lib.grid_with_mode(v._grid, self._mode)
        cfg.grid = self._grid
        cfg.max_points, cfg.out_capacity, cfg.send_capacity = self.max_points, cap, self.send_cap
        cfg.pointnet_pack = pointnet.pointnet_pack.data_ptr()
        cfg.enc_ws, cfg.enc_ws_bytes, cfg.enc_ws_max_points = self._enc_ws.data_ptr(), need, self.max_points
        cfg.max_depth = self.max_depth
        if tsdf_vol is not None:
            t = tsdf_vol
            cfg.tsdf.tsdf, cfg.tsdf.weight, cfg.tsdf.color = t.tsdf.data_ptr(), t.weight.data_ptr(), t.color.data_ptr()
            for a in range(3):
                cfg.tsdf.dim[a] = int(t._vol_dim[a])
                cfg.tsdf.origin[a] = float(t._vol_origin[a])
            cfg.tsdf.voxel_size = float(np.float32(t._voxel_size))
            cfg.tsdf.trunc_margin = float(np.float32(t._trunc_margin))
        cfg.n_slots = S
        for s in range(S):
            b = cfg.slots[s]
            b.input_pts, b.feats, b.pcounts = self.input_pts[s].data_ptr(), self.feats[s].data_ptr(), self.pcounts[s].data_ptr()
            b.flat_ids, b.grid_ids, b.counters = self.flat_ids[s].data_ptr(), self.grid_ids[s].data_ptr(), self.counters[s].data_ptr()
            b.sdf = self.sdf[s].data_ptr()
            b.send_block = self.send[s].data_ptr() if self.send is not None else None
            b.host_words = self.host[s].data_ptr()
        cfg.encode_stream, cfg.main_stream = self.enc.cuda_stream, self.main.cuda_stream
        if self.double_buffered:
            cfg.enc_ws2 = self._enc_ws2.data_ptr()
            cfg.front_stream, cfg.blend_stream = self.front.cuda_stream, self.blend.cuda_stream
        cfg.encoder_workgroups = self.encoder_workgroups
        self._cfg = cfg
        h = C.c_void_p()
        _lib.check(lib.bnv_frame_pipe_create(C.byref(cfg), C.byref(h)), "bnv_frame_pipe_create")
        self._h = h
        self._next = 0
        self._busy = [False] * S
        self._reserved = [0] * S
        self._decode = [False] * S
        self._epoch = [0] * S
        self._lws = [None] * S
        self._lws_next = 0           # decode workspace of the next frame upserted (two alternate with a blend stream)
        self._lws_generation = v._lws_generation
        # persistent lattice tables (include/bnv_fusion.h: bnv_volume_t.lattice_table): SDF table entries of rows a
        # frame did not update are carried over instead of re-evaluated (~5 % of a frame's entries in a steady scan).
        # ``persistent_tables=False`` / BNV_PERSISTENT_TABLES=0 switches it off (every entry recomputed each frame).
        if persistent_tables is None:
            persistent_tables = os.environ.get("BNV_PERSISTENT_TABLES", "1") != "0"
        self.persistent_tables = bool(persistent_tables)
        if self.persistent_tables:
            if v._phave is not None:
                v.invalidate_tables()      # tables another pipe (another model's networks, perhaps) left on this volume
            v.enable_persistent_tables()
        # what the carried-over entries were computed with: (arithmetic mode, the SDF network's pack version)
        self._tables_key = None
        self._slot_mode = [None] * self.n_slots
        self.n_lattice_ws = 2
        self._words = (C.c_int32 * HOST_WORDS)()
        self._keep = [None] * S
        self.inputs_resident = False
        torch.cuda.synchronize(dev)      # the zero-filled buffers above are complete before any other stream uses them

    @property
    def tsdf_stream(self):
        """The stream the TSDF side fusion of a frame runs on (a synchronous TSDF update on another stream must be
        ordered before it): the encode stream (csrc/pipeline.hip)."""
        return self.enc

    def close(self):
        """Destroys the C object (every frame must have been collected)."""
        if getattr(self, "_h", None):
            self._lib.bnv_frame_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check_device(self, t, name):
        if not t.is_cuda or (self.dev.index is not None and t.device.index != self.dev.index):
            raise _lib.BnvError(f"FramePipe: {name} is on {t.device}, the pipe runs on {self.dev} "
                                "(frames are device tensors; there is no host path)")

    # ---- phases -----------------------------------------------------------------------------------
    def free_slot(self):
        """A slot no uncollected frame holds -- the next of the ring if it is free, else the first free one behind it
        (frames may be collected in any order) -- or None."""
        for k in range(self.n_slots):
            s = (self._next + k) % self.n_slots
            if not self._busy[s]:
                return s
        return None

    def begin(self, frame, slot=None):
        s = self.free_slot() if slot is None else slot
        if s is None or self._busy[s]:
            raise _lib.BnvError("FramePipe.begin: every slot holds an uncollected frame (collect one with result())")
        lib = self._lib
        # conversions first (they run on the CALLER's stream), then the encode stream waits for that stream: whenever a
        # conversion really ran -- or the caller has not declared its frames complete in device memory
        # (inputs_resident) -- the encode must not start before the caller's stream has produced the buffer
        col = None
        converted = False
        # the TSDF side fusion reads the frame's depth image whether or not the frame also carries input_pts (the
        # reference dataset's frames hold both, run_e2e.py:78-109)
        side = self.tsdf_vol is not None and frame.get("depth") is not None
        if side and frame.get("rgb") is not None:
            col = self.tsdf_vol._fold_color(frame["rgb"])                 # (caller's stream)
            converted = True
        side_d = None
        if "input_pts" in frame:
            src = frame["input_pts"]
            self._check_device(src, "input_pts")
            pts = src[0].detach().float().contiguous()
            converted |= pts.data_ptr() != src.data_ptr() or pts.dtype != src.dtype
            if side:
                sd = torch.as_tensor(frame["depth"])
                if sd.dtype in (torch.uint16, torch.int16):
                    side_d = sd.to(self.dev).contiguous()
                else:
                    side_d = sd.to(self.dev, torch.float32).contiguous()
                converted |= (not sd.is_cuda) or side_d.data_ptr() != sd.data_ptr()
        else:
            src = frame["depth"]
            self._check_device(src, "depth")
            d = src.contiguous()
            converted |= d.data_ptr() != src.data_ptr()
            if d.dtype == torch.float64 and self.tsdf_vol is not None:
                raise _lib.BnvError("FramePipe with a TSDF side volume takes uint16 (mm) or float32 (m) depth images")
        if not self.inputs_resident or converted:
            (self.front or self.enc).wait_stream(self.main)     # the stream the frame's first kernel runs on
        mode = _lib.model_mode(self.pointnet)            # the frame keeps it through its decode (per slot, in C)
        if mode != self._mode:
            _lib.check(lib.bnv_frame_pipe_set_mlp_mode(self._h, mode + 1), "bnv_frame_pipe_set_mlp_mode")
            self._mode = mode
        if "input_pts" in frame:
            self._keep[s] = (pts, side_d, col)                            # alive until the slot is begun again
            _lib.check(lib.bnv_frame_begin_points(self._h, s, _lib.ptr(pts), int(pts.shape[0])), "bnv_frame_begin_points")
            if side_d is not None:
                H, W = int(side_d.shape[-2]), int(side_d.shape[-1])
                K = (C.c_double * 9)(*np.asarray(frame["intr_mat"], dtype=np.float64)[:3, :3].reshape(-1))
                T = (C.c_double * 16)(*np.asarray(frame["T_wc"], dtype=np.float64).reshape(-1))
                _lib.check(lib.bnv_frame_side_depth(self._h, s, _lib.ptr(side_d), self._dtypes[side_d.dtype], H, W, K, T,
                                                    _lib.ptr(col)), "bnv_frame_side_depth")
        else:
            H, W = int(d.shape[-2]), int(d.shape[-1])
            K = (C.c_double * 9)(*np.asarray(frame["intr_mat"], dtype=np.float64)[:3, :3].reshape(-1))
            T = (C.c_double * 16)(*np.asarray(frame["T_wc"], dtype=np.float64).reshape(-1))
            self._keep[s] = (d, col)
            _lib.check(lib.bnv_frame_begin_depth(self._h, s, _lib.ptr(d), self._dtypes[d.dtype], H, W, K, T,
                                                 _lib.ptr(col)), "bnv_frame_begin_depth")
        self._busy[s] = True
        self._slot_mode[s] = mode
        self._next = (s + 1) % self.n_slots
        return s

    def bound(self, slot):
        m = C.c_int32(0)
        _lib.check(self._lib.bnv_frame_bound(self._h, slot, C.byref(m)), "bnv_frame_bound")
        return int(m.value)

    def upsert(self, slot, decode=True, ghost_rows=0):
        """Upsert of the slot's encoded voxels; ``ghost_rows``: rows the frame's install may create on top (the
        volume is grown for both BEFORE the upsert: the decode-origin stamps live in a workspace that growth
        re-makes).  Returns the slot's send block (int32 words) or None."""
        v = self.volume
        need = self.cap + int(ghost_rows)
        v._reserve(need)
        v._rows_upper += need
        v._inflight += need
        self._reserved[slot] = need
        self._decode[slot] = bool(decode)
        lws = None
        if decode:
            lws, self._epoch[slot] = v._lattice_workspace(self.cap, self._lws_next if self.double_buffered else 0)
            self._lws_next = (self._lws_next + 1) % self.n_lattice_ws
            if self._lws_generation != v._lws_generation:
                # the volume has re-made its decode workspaces (it grew): the pointers the C object remembers are gone
                _lib.check(self._lib.bnv_frame_pipe_forget_workspaces(self._h), "bnv_frame_pipe_forget_workspaces")
                self._lws_generation = v._lws_generation
        self._lws[slot] = lws
        ws = v._workspace(self.cap)
        _lib.check(self._lib.bnv_frame_upsert(self._h, slot, C.byref(v._struct()), _lib.ptr(ws), ws.numel(),
                                              _lib.ptr(lws), self._epoch[slot]), "bnv_frame_upsert")
        return None if self.send is None else self.send[slot]

    def _vstruct(self):
        """The volume as the pipe's calls see it: with the persistent lattice tables switched on for them."""
        s = self.volume._struct()
        if self.persistent_tables and self.volume._phave is not None:
            s.lattice_persist = 1
        return s

    def finish(self, slot, blocks=None, capacity=0):
        v = self.volume
        nerf = self.pointnet.nerf
        lws = self._lws[slot]
        d, keep = v._delta(self.sdf_delta)
        key = (self._slot_mode[slot], getattr(nerf, "pack_version", 0))
        if self.persistent_tables and lws is not None and key != self._tables_key:
            # table entries are carried across frames: entries computed in another arithmetic mode, or with other SDF
            # weights (load_state_dict -> repack() rewrites nerf.sdf_pack in place on the same model object), must not
            # be.  On the main stream, in front of this frame's marking
            if self._tables_key is not None:
                v.invalidate_tables()
            self._tables_key = key
        _lib.check(self._lib.bnv_frame_finish(self._h, slot, C.byref(self._vstruct()), _lib.ptr(blocks), int(capacity),
                                              _lib.ptr(nerf.sdf_pack), C.byref(d), _lib.ptr(lws),
                                              lws.numel() if lws is not None else 0, self._epoch[slot]),
                   "bnv_frame_finish")

    def cancel(self, slot):
        """Abandons a frame that was begun and not upserted (include/bnv_fusion.h: bnv_frame_cancel); the slot is free."""
        _lib.check(self._lib.bnv_frame_cancel(self._h, slot), "bnv_frame_cancel")
        self._busy[slot] = False

    def ready(self, slot):
        r = self._lib.bnv_frame_ready(self._h, slot)
        if r < 0:
            _lib.check(r, "bnv_frame_ready")
        return bool(r)

    def result(self, slot):
        """Host wait for the slot's frame -> its pinned words as a numpy int32 array (a copy); settles the frame's row
        reservation and raises on device-side errors.  The slot is free again afterwards."""
        _lib.check(self._lib.bnv_frame_result(self._h, slot, self._words), "bnv_frame_result")
        w = np.frombuffer(self._words, dtype=np.int32).copy()
        self._busy[slot] = False
        v = self.volume
        v.settle(self._reserved[slot], int(w[W_STATUS]))
        v.check_status(int(w[W_STATUS + 1]))
        if int(w[W_COUNTERS + 4]):
            from .fusion import encode_error_message
            raise _lib.BnvError(encode_error_message(int(w[W_COUNTERS + 4])))
        return w

    def outputs(self, slot, words, copy=True):
        """(coords [U', 3] i64, sdf [U', 27] or None) of a collected frame, or (None, None) for a frame without a point
        inside the volume.  ``copy=False``: views into the slot's buffers, valid until the slot is begun again."""
        if int(words[W_COUNTERS]) == 0:
            return None, None
        self.volume.track_n_pts(float(words[W_COUNTERS + 3: W_COUNTERS + 4].view(np.float32)[0]))
        n_out = int(words[W_COUNTERS + 2])
        c = self.grid_ids[slot, :n_out]
        s = self.sdf[slot, :n_out] if self._decode[slot] else None
        if copy:
            # The copies run on the stream the frame's last kernels ran on (they are through: result() has waited),
            # not on the caller's, which may hold several later frames' work; and the slot -- free again for begin() --
            # is not written before they are done: the stream of a frame's FIRST kernel waits for them.  (Cloning on
            # the caller's stream let the next frame's encode, a frame or two ahead on its own streams, overwrite the
            # slot first.)
            cur = torch.cuda.current_stream(self.dev)
            side = self.blend or self.main
            with torch.cuda.stream(side):
                c = c.clone()
                s = None if s is None else s.clone()
                ev = torch.cuda.Event()
                ev.record(side)
            (self.front or self.enc).wait_event(ev)
            if side.cuda_stream != cur.cuda_stream:
                cur.wait_event(ev)
                c.record_stream(cur)
                if s is not None:
                    s.record_stream(cur)
        return c, s
