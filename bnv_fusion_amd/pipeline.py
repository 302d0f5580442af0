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

Sharded, early exchange (opt-in, BNV_EARLY_EXCHANGE=1; csrc/shard.hip): begin -> bound -> ``send = pipe.exchange_begin(slot, capacity)``
-> all-gather on ``pipe.exchange_stream()`` -> ``pipe.exchange_end(slot)`` -> upsert -> finish(slot, blocks, capacity).
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
    # stream's chain (upsert -> exchange -> install -> mark) runs beside it; all CUs otherwise.  BNV_PIPE_ENCODER_WGS
    # overrides (0 = all CUs).
    ENCODER_SHARE_SHARDED = 0.75
    # Five streams with CU-masked encoder and table streams (csrc/pipeline.hip): CUs of the table kernel, CUs of the
    # encoder; the rest belongs to nobody (the chain's 1,024-thread kernels cannot share a CU with either MLP kernel).
    # BNV_PIPE_CU_SPLIT="table,encoder" overrides; "0" = the four-stream schedule.
    # Measured (profiles/r04_cu_mask_experiment.txt): the partition works, but a rank's frame gets SLOWER (0.29-0.33
    # against 0.263 ms at world 8): with both MLP kernels resident all the time the chain's 1,024-thread kernels find
    # too few CUs.  Opt-in therefore (cu_split=(160, 64), multiples of 32: an equal number of CUs per shader engine --
    # other counts leave workgroups of the persistent kernels waiting for a CU and double their time).
    CU_SPLIT_SHARDED = None

    def __init__(self, volume, pointnet, max_points, n_slots=4, tsdf_vol=None, max_depth=3.0, sdf_delta=None,
                 streams=4, encoder_workgroups=None, cu_split=None, exchange_stream=True):
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
        from .streams import concurrent_stream
        import os
        self.main = torch.cuda.current_stream(dev)
        self._masked = []                                     # CU-masked streams this pipe created (destroyed with it)
        split = os.environ.get("BNV_PIPE_CU_SPLIT")
        if split is not None:
            split = tuple(int(x) for x in split.split(",")) if split not in ("", "0") else None
        elif cu_split is not None:
            split = tuple(cu_split) if cu_split else None
        else:
            split = self.CU_SPLIT_SHARDED if (self.world > 1 and int(os.environ.get("BNV_PIPE_STREAMS", streams)) >= 4
                                              and _lib.model_mode(pointnet) != 2) else None
        cus = int(lib.bnv_num_compute_units())
        if split is not None and (len(split) != 2 or min(split) < 1 or sum(split) > cus):
            raise ValueError(f"FramePipe: CU split {split} does not fit {cus} CUs")
        self.cu_split = split
        # hipExtStreamCreateWithCUMask makes BLOCKING streams: every operation on the legacy default stream (torch's
        # default "current stream") waits for them and holds them back in turn (measured: 1.2 ms per frame instead of
        # 0.27).  The pipe then runs on a main stream of its own; callers enter it through stream_context() (the
        # sharded backend does), and nothing of a frame may touch the default stream.
        # BNV_PIPE_MAIN_HIGH=1 (experiment): a HIGH-PRIORITY main stream of the pipe's own without CU masks -- the chain on it
        # (upsert .. table) is a rank's critical path, the front end / encoder of later frames run ahead on the others
        high = os.environ.get("BNV_PIPE_MAIN_HIGH", "0") == "1"
        self.own_main = (split is not None or high) and self.main.cuda_stream == 0
        if self.own_main:
            # (high priority: the chain on it is the frame's critical path; front end and blend run ahead / behind)
            self.main = torch.cuda.Stream(device=dev, priority=int(os.environ.get("BNV_PIPE_MAIN_PRIORITY", -1)))
        if split is not None:
            # table kernel on CUs [0, t), encoder on [t, t + e) of the device's enumeration (which interleaves the
            # XCDs: each set takes an equal share of every XCD)
            self.table = self._masked_stream(range(0, split[0]), cus)
            self.enc = self._masked_stream(range(split[0], split[0] + split[1]), cus)
            encoder_workgroups = split[1]
        else:
            self.enc = concurrent_stream(dev, self.main)      # verified to overlap the main stream
        # the front end (voxelise + rank) and the blend on streams of their own (csrc/pipeline.hip: why four)
        streams = int(os.environ.get("BNV_PIPE_STREAMS", streams))
        self.front = self.blend = None
        if split is None:
            self.table = None
        if streams >= 4 or split is not None:
            self.front = concurrent_stream(dev, self.main, exclude=(self.enc,))
            self.blend = concurrent_stream(dev, self.main, exclude=(self.enc, self.front))
        # streams = 5 WITHOUT masks (BNV_PIPE_STREAMS=5, BNV_PIPE_CU_SPLIT=0): the table MLP on an ordinary stream of
        # its own.  Correct but slower than four streams: with both MLP kernels in flight all the time and nothing
        # reserving CUs, the chain's small kernels crawl (profiles/r04_fifth_stream_experiment.txt).
        if streams >= 5 and split is None:
            self.table = concurrent_stream(dev, self.main, exclude=(self.enc, self.front, self.blend))
        self.double_buffered = self.front is not None
        # early exchange (csrc/shard.hip, csrc/pipeline.hip; OPT-IN: BNV_EARLY_EXCHANGE=1): a sharded frame's records
        # carry its CONTRIBUTION to the boundary voxels and leave behind the encode; the all-gather runs on `xchg`, a
        # stream of its own, while the main stream still decodes the frame before.  Bit-identical results; it takes the
        # collective's latency off the main stream's chain.  Priced on one GPU it is SLOWER (0.37 against 0.31 ms per
        # frame for a rank of 8, profiles/r05_experiments.txt [e8]): the table kernel excludes every other kernel, so
        # each stream's work of a frame has to fit into the window between two table kernels, and the encode stream's
        # (encoder + finalize + emit, ~120 us) is as long as the main stream's chain WITH the exchange in it -- shortening
        # that chain only closes the window earlier.  Default therefore: records of the rows after the upsert,
        # all-gather on the main stream.  BNV_EXCHANGE_STREAM=0 / exchange_stream=False: early records, but the
        # all-gather on the main stream (in-process drivers that run several shards in lock step on one stream).
        self.early_exchange = (self.world > 1 and self.table is None
                               and os.environ.get("BNV_EARLY_EXCHANGE", "0") == "1")
        self.xchg = None
        if self.early_exchange and exchange_stream and os.environ.get("BNV_EXCHANGE_STREAM", "1") != "0":
            others = tuple(x for x in (self.enc, self.front, self.blend) if x is not None)
            self.xchg = concurrent_stream(dev, self.main, exclude=others)
            if not self.xchg.bnv_concurrent and self.front is not None:
                # four hardware queues by default (GPU_MAX_HW_QUEUES): a fifth stream shares one.  Then with the front
                # stream, which runs a frame or two ahead of everything else
                self.xchg = concurrent_stream(dev, self.main, exclude=tuple(x for x in others if x is not self.front))
        if encoder_workgroups is None:
            encoder_workgroups = os.environ.get("BNV_PIPE_ENCODER_WGS")
            if encoder_workgroups is None:
                cus = int(lib.bnv_num_compute_units())
                encoder_workgroups = int(cus * self.ENCODER_SHARE_SHARDED) if (self.world > 1 and streams >= 4) else 0
        self.encoder_workgroups = int(encoder_workgroups)
        # five streams: both MLP kernels are in flight all the time and their workgroup counts partition the CUs (the
        # encoder's and the table kernel's share of a frame's MLP work, a few CUs left to the small kernels)
        self.table_workgroups = split[0] if split is not None else int(os.environ.get("BNV_PIPE_TABLE_WGS", 0))
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
            if self.table is not None:
                cfg.table_stream = self.table.cuda_stream
                cfg.table_workgroups = self.table_workgroups
        cfg.encoder_workgroups = self.encoder_workgroups
        cfg.early_exchange = int(self.early_exchange)
        # BNV_PIPE_ENCODER_GATE=k (experiment; csrc/pipeline.hip): the encoder of a frame starts behind the table kernel of
        # the k-th frame before it.  Measured with the early exchange (k = 2, 3): no better than ungated.
        self.encoder_gate = int(os.environ.get("BNV_PIPE_ENCODER_GATE", 0))
        cfg.encoder_gate = self.encoder_gate
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
        # Not with the snapshot schedule (table stream); BNV_PERSISTENT_TABLES=0 switches it off.
        self.persistent_tables = self.table is None and os.environ.get("BNV_PERSISTENT_TABLES", "1") != "0"
        if self.persistent_tables:
            if v._phave is not None:
                v.invalidate_tables()      # tables another pipe (another model's networks, perhaps) left on this volume
            v.enable_persistent_tables()
        self._tables_mode = None
        self._slot_mode = [None] * self.n_slots
        # with a table stream, three: a frame's upsert (which stamps into the workspace) then waits for the blend of
        # the frame THREE back, not two -- with two the chain of frame t+2 could only start behind table(t) + blend(t)
        # and the table stream idled for the rest of that chain
        self.n_lattice_ws = int(os.environ.get("BNV_PIPE_LATTICE_WS", 3 if self.table is not None else 2))
        self._words = (C.c_int32 * HOST_WORDS)()
        self._keep = [None] * S
        self.inputs_resident = False
        torch.cuda.synchronize(dev)      # the zero-filled buffers above are complete before any other stream uses them

    def stream_context(self):
        """Context in which the caller drives the pipe (and everything between its phases): the pipe's own main stream
        when it has one (CU-masked streams, see __init__), nothing otherwise."""
        import contextlib
        return torch.cuda.stream(self.main) if self.own_main else contextlib.nullcontext()

    @property
    def tsdf_stream(self):
        """The stream the TSDF side fusion of a frame runs on (a synchronous TSDF update on another stream must be
        ordered before it): the blend stream with a table stream, the encode stream otherwise (csrc/pipeline.hip)."""
        return self.blend if self.table is not None else self.enc

    def _masked_stream(self, cu_ids, cus):
        words = (cus + 31) // 32
        mask = (C.c_uint32 * words)()
        for i in cu_ids:
            mask[i // 32] |= 1 << (i % 32)
        out = C.c_void_p()
        with torch.cuda.device(self.dev):
            _lib.check(self._lib.bnv_stream_create_cu_mask(words, mask, C.byref(out)), "bnv_stream_create_cu_mask")
        self._masked.append(out.value)
        st = torch.cuda.ExternalStream(out.value, device=self.dev)
        st.bnv_concurrent = True          # a queue of its own
        return st

    def close(self):
        """Destroys the C object and the CU-masked streams (every frame must have been collected)."""
        if getattr(self, "_h", None):
            self._lib.bnv_frame_pipe_destroy(self._h)
            self._h = None
        for h in getattr(self, "_masked", []):
            torch.cuda.synchronize(self.dev)
            self._lib.bnv_stream_destroy(C.c_void_p(h))
        self._masked = []

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

    def exchange_stream(self):
        """The stream the caller's all-gather of an early exchange runs on (the main stream without one of its own)."""
        return self.xchg if self.xchg is not None else torch.cuda.current_stream(self.dev)

    def exchange_begin(self, slot, capacity):
        """Early exchange: orders the exchange stream behind the slot's encode -> the slot's send block (header +
        ``capacity`` records, int32 words) to all-gather ON THAT STREAM."""
        if (capacity + 1) * REC_WORDS > self.send[slot].numel():
            raise _lib.BnvError(f"exchange capacity {capacity} exceeds the slot's send block "
                                f"({self.send[slot].numel() // REC_WORDS - 1} records)")
        _lib.check(self._lib.bnv_frame_exchange_begin(self._h, slot, C.c_void_p(self.exchange_stream().cuda_stream)),
                   "bnv_frame_exchange_begin")
        return self.send[slot][: (capacity + 1) * REC_WORDS]

    def exchange_end(self, slot):
        """The slot's all-gather is enqueued on the exchange stream: finish() orders the main stream behind it."""
        _lib.check(self._lib.bnv_frame_exchange_end(self._h, slot, C.c_void_p(self.exchange_stream().cuda_stream)),
                   "bnv_frame_exchange_end")

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
            lws, self._epoch[slot] = v._lattice_workspace(self.cap, self._lws_next if self.double_buffered else 0,
                                                          snapshot=self.table is not None)
            self._lws_next = (self._lws_next + 1) % self.n_lattice_ws
            if self._lws_generation != v._lws_generation:
                # the volume has re-made its decode workspaces (it grew): the pointers the C object remembers are gone
                _lib.check(self._lib.bnv_frame_pipe_forget_workspaces(self._h), "bnv_frame_pipe_forget_workspaces")
                self._lws_generation = v._lws_generation
        self._lws[slot] = lws
        ws = v._workspace(self.cap)
        _lib.check(self._lib.bnv_frame_upsert(self._h, slot, C.byref(v._struct()), _lib.ptr(ws), ws.numel(),
                                              _lib.ptr(lws), self._epoch[slot]), "bnv_frame_upsert")
        return None if (self.send is None or self.early_exchange) else self.send[slot]

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
        if self.persistent_tables and lws is not None and (self._slot_mode[slot] != self._tables_mode or v._tables_dirty):
            # table entries are carried across frames: entries computed in another arithmetic mode (or from features
            # somebody wrote behind the library's back) must not be.  On the main stream, in front of this frame's marking
            if self._tables_mode is not None or v._tables_dirty:
                v.invalidate_tables()
            self._tables_mode = self._slot_mode[slot]
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
