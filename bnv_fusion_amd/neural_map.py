"""NeuralMap -- the driver object of the reference (src/run_e2e.py:27-194) on the HIP path:

* ``integrate(frame)``            run_e2e.py:78-109: encode -> track_n_pts -> _integrate (+ TSDF side fusion);
* ``fuse_and_decode[_async]``     one unit of the benchmark metric ("depth frames/sec fused+decoded", SURVEY.md
                                  section 8d): integrate + the 3x3x3 meshing lattice of every voxel the frame's
                                  encode returned; the async form never synchronises with the host;
* ``optimize``                    run_e2e.py:111-162: the global optimiser over ``self.frames``;
* ``extract_mesh`` / ``save``     run_e2e.py:164-194.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import BnvError as _lib_error
from .sparse_volume import SparseVolume


def frame_input_pts(frame, max_depth=3.0):
    """``frame['input_pts']`` if present (the reference's dataset output), else built on the GPU from
    ``frame['depth']`` (uint16 mm or float metres), ``frame['intr_mat']`` and ``frame['T_wc']`` by the
    front-end kernel (csrc/frontend.hip) -- without a host read: invalid pixels are NaN rows that the
    encoder's bounds mask drops."""
    if "input_pts" in frame:
        return frame["input_pts"]
    from .frontend import depth_to_input_pts
    pts, _ = depth_to_input_pts(frame["depth"], frame["intr_mat"], frame["T_wc"], max_depth=max_depth, compact=False)
    return pts


class FrameHandle:
    """A frame enqueued by NeuralMap.fuse_and_decode_async.  ``result()`` waits for THAT frame only (a HIP
    event recorded behind its last kernel) and returns (coords [U',3] i64, sdf [U',27] or None)."""

    def __init__(self, nm, bufs, host_counters, event, cap, sdf, host_rows=None):
        self._nm, self._bufs, self._host, self._event, self._cap, self._sdf = nm, bufs, host_counters, event, cap, sdf
        self._host_rows = host_rows
        self._done = None
        self._settled = False

    def __del__(self):
        # a handle dropped without result(): give its row reservation back (the bound stays an upper bound)
        if not self._settled and self._host_rows is not None:
            try:
                self._nm.volume.release(self._cap)
            except Exception:
                pass

    def result(self):
        if self._done is not None:
            return self._done
        self._event.synchronize()
        h = self._host
        n_valid, n_out, err = int(h[0]), int(h[2]), int(h[4])
        vol = self._nm.volume
        # the reservation was made for the capacity bound; the row count behind this frame's integrate came back
        # with the counters, so the host-side bound stays exact and _reserve never has to synchronise
        vol.settle(self._cap, int(self._host_rows[0]))
        self._settled = True
        vol.check_status(self._host_rows[1])      # sticky error word of the upsert kernels, read back with the rows
        if err:
            from .fusion import encode_error_message
            raise _lib_error(encode_error_message(err))
        if n_valid == 0:
            self._done = (None, None)
        else:
            vol.track_n_pts(float(h[3:4].view(torch.float32)[0]))
            feats, pcounts, flat_ids, grid_ids = self._bufs
            self._done = (grid_ids[:n_out], None if self._sdf is None else self._sdf[:n_out])
        self._bufs = None
        return self._done


class PipeHandle:
    """A frame enqueued through the C frame pipeline (pipeline.FramePipe).  ``result()`` waits for that frame only."""

    def __init__(self, nm, pipe, slot):
        self._nm, self._pipe, self._slot, self._done, self._err = nm, pipe, slot, None, None

    @property
    def pending(self):
        return self._done is None and self._err is None

    def result(self):
        if self._err is not None:
            raise self._err
        if self._done is None:
            try:
                w = self._pipe.result(self._slot)
                self._done = self._pipe.outputs(self._slot, w, copy=self._nm.copy_results)
            except Exception as e:
                self._err = e
                raise
        return self._done


class NeuralMap:
    def __init__(self, dimensions, voxel_size, pointnet, min_pts_in_grid=8, feature_vector_size=8,
                 capacity=100000, device="cuda:0", tsdf=False, truncated_units=10, sdf_delta_weight=0.1,
                 max_depth=3.0):
        self.pointnet = pointnet
        # depth cut-off of the reference's loader (model.ray_tracer.ray_max_dist, fusion_pointnet_model.yaml:43 ->
        # fusion_inference_dataset.py:28 -> common.py:110-113): applies to input_pts AND to the TSDF side fusion,
        # both of which see ``depth * mask`` in the reference (run_e2e.py:100-109 reads frame['rgbd'])
        self.max_depth = max_depth
        self.volume = SparseVolume(feature_vector_size, voxel_size, dimensions, min_pts_in_grid,
                                   capacity=capacity, device=device)
        self.voxel_size = voxel_size
        self.dimensions = dimensions
        self.sdf_delta = None
        self.tsdf_vol = None
        self.tsdf_voxel_size = 0.025                                      # run_e2e.py:58
        self.truncated_units = truncated_units
        self.truncated_dist = min(truncated_units * voxel_size * 0.5, 0.1)  # run_e2e.py:54
        self.frames = []                                                  # key frames for optimize() (run_e2e.py:73)
        # fuse_and_decode_async: the encode of a frame depends only on the frame, so it is enqueued on a second
        # HIP stream and overlaps the previous frame's integrate / decode kernels on the main stream
        self.overlap_encode = True
        self._enc_stream = None
        # fuse_and_decode_async through the C frame pipeline (csrc/pipeline.hip: persistent slots, four streams -- front
        # end / encoder / upsert + decode tables / blend -- no per-frame tensor, event or pinned allocation); False:
        # the per-stage calls of rounds 1-3 on two streams.  Bit-identical results either way.
        self.frame_pipe = True
        self.copy_results = True      # False: results are views into the pipeline's slots (valid for 2 more frames)
        self._pipe = None
        self._pipe_open = []
        # True: the caller guarantees that a frame's tensors are complete in device memory when it is passed in
        # (uploaded / produced and synchronised earlier), so the encode stream need not wait for the caller's stream
        self.inputs_resident = False
        self._vol_ev = None            # behind the last TSDF update a synchronous integrate() enqueued (main stream)
        self.sdf_delta_weight = sdf_delta_weight                          # fusion_pointnet_model.yaml:44,47
        if tsdf:                                                          # run_e2e.py:60-71
            import numpy as np
            from .sparse_volume import get_world_range
            from .tsdf import TSDFVolume
            mn, mx, _ = get_world_range(dimensions, self.tsdf_voxel_size)
            self.tsdf_vol = TSDFVolume(np.stack([mn, mx], 1), self.tsdf_voxel_size, device=device)

    def prepare_tsdf_volume(self):
        """run_e2e.py:169-186 -> sdf_delta [1, 1, X, Y, Z] for decode_pts / meshlize."""
        return self.tsdf_vol.sdf_delta(self.truncated_dist, self.sdf_delta_weight)

    def _integrate_tsdf(self, frame, gate=None):
        """run_e2e.py:99-109: TSDF side fusion of the same frame.  ``gate``: device int32 (the frame's in-bounds
        point count): nothing happens when it is 0, as in the reference, which returns before this call then."""
        if self.tsdf_vol is None or "depth" not in frame:
            return
        # uint16 millimetres go to the kernel as they are (converted per sample, no float copy of the frame)
        self.tsdf_vol.integrate(frame.get("rgb"), frame["depth"], frame["intr_mat"], frame["T_wc"], obs_weight=1.,
                                max_depth=self.max_depth, gate=gate)

    def _drain_pipe(self):
        """Collects every frame still in the frame pipeline (their side streams also write the TSDF volume)."""
        while self._pipe_open:
            h = self._pipe_open.pop(0)
            if h.pending:
                h.result()

    def _pipe_frame(self, frame, decode):
        from .pipeline import FramePipe
        n = int(frame["input_pts"].shape[1]) if "input_pts" in frame else int(frame["depth"].shape[-2] * frame["depth"].shape[-1])
        if self._pipe is None or self._pipe.max_points < n or self._pipe.pointnet is not self.pointnet \
                or self._pipe.tsdf_vol is not self.tsdf_vol:
            self._drain_pipe()
            self._pipe = FramePipe(self.volume, self.pointnet, n, n_slots=4, tsdf_vol=self.tsdf_vol,
                                   max_depth=self.max_depth)
        pipe = self._pipe
        pipe.inputs_resident, pipe.sdf_delta = self.inputs_resident, self.sdf_delta
        self._pipe_open = [h for h in self._pipe_open if h.pending]
        while pipe.free_slot() is None:                 # every slot holds an uncollected frame: collect the oldest
            self._pipe_open.pop(0).result()
        if self._vol_ev is not None:                    # a synchronous integrate()'s TSDF update on the caller's stream
            pipe.tsdf_stream.wait_event(self._vol_ev)
            self._vol_ev = None
        with torch.no_grad():
            slot = pipe.begin(frame)
            pipe.upsert(slot, decode=decode)
            pipe.finish(slot)
        h = PipeHandle(self, pipe, slot)
        self._pipe_open.append(h)
        return h

    def integrate(self, frame):
        """run_e2e.py:78-98.  frame['input_pts'] : [1, N, 6] float32 on the GPU (or a depth frame, see
        frame_input_pts).
        Returns the voxel coordinates the frame touched ([U', 3] int64) or None."""
        self._drain_pipe()
        input_pts = frame_input_pts(frame, self.max_depth)
        if len(input_pts) == 0:
            return None
        with torch.no_grad():
            fine_feats, fine_weights, _, fine_coords, fine_n_pts = self.pointnet.encode_pointcloud(
                input_pts, self.volume.n_xyz, self.volume.min_coords, self.volume.max_coords,
                self.volume.voxel_size, return_dense=self.pointnet.dense_volume)
            if fine_feats is None:
                return None
            self.volume.track_n_pts(fine_n_pts)
            self.pointnet._integrate(self.volume, fine_coords, fine_feats, fine_weights)
            self._integrate_tsdf(frame)
            self._mark_volume_update()
        return fine_coords

    def _mark_volume_update(self):
        """An event behind a TSDF update enqueued on the current stream: the encode stream of a later
        fuse_and_decode_async (which also updates the TSDF volume) waits for it."""
        if self.tsdf_vol is not None:
            self._vol_ev = torch.cuda.Event()
            self._vol_ev.record()

    def fuse_and_decode(self, frame):
        """One benchmark work unit: integrate + SDF lattice [U', 27] of the touched voxels (live
        volume values, i.e. decode_pts(..., is_coords=True, query_tensor=False))."""
        coords = self.integrate(frame)
        if coords is None:
            return None, None
        sdf = self.volume.decode_lattice(coords, self.pointnet.nerf, self.sdf_delta, query_tensor=False)
        return coords, sdf

    def _encode_frame_async(self, frame):
        """encode_pointcloud of a frame without a host sync: from ``input_pts`` if the frame carries them (the
        reference's dataset output), else straight from its depth image (front end fused into the voxelisation)."""
        v = self.volume
        if "input_pts" in frame:
            return self.pointnet.encode_pointcloud_async(frame["input_pts"], v.n_xyz, v.min_coords, v.max_coords,
                                                         v.voxel_size)
        return self.pointnet.encode_depth_async(frame["depth"], frame["intr_mat"], frame["T_wc"], self.max_depth,
                                                v.n_xyz, v.min_coords, v.max_coords, v.voxel_size)[:6]

    def fuse_and_decode_async(self, frame, decode=True):
        """The same work as fuse_and_decode, enqueued without any host synchronisation: integrate and the
        lattice decode read the frame's voxel count from device memory.  Returns a FrameHandle; call
        ``.result()`` after enqueuing the NEXT frame -- or the next TWO: then the encode of frame t + 2 is always
        queued before frame t + 1's upsert / decode front end runs (+5 % frames/s, bench.py) -- to keep the GPU busy
        (``result()`` also settles the frame's row reservation and raises on a device-side upsert error).  Any number
        of uncollected frames gives the synchronous results bit for bit."""
        if self.frame_pipe and self.overlap_encode and not self.pointnet.dense_volume and not (
                "input_pts" not in frame and frame["depth"].dtype == torch.float64 and self.tsdf_vol is not None):
            return self._pipe_frame(frame, decode)
        self._drain_pipe()
        with torch.no_grad():
            v = self.volume
            main = torch.cuda.current_stream()
            if self.overlap_encode:
                if self._enc_stream is None:
                    from .streams import concurrent_stream
                    self._enc_stream = concurrent_stream(v._dev, main)     # one that really runs beside `main`
                enc = self._enc_stream
                if self.inputs_resident:
                    # only a synchronous integrate() on the caller's stream can still hold the TSDF volume
                    if self._vol_ev is not None:
                        enc.wait_event(self._vol_ev)
                        self._vol_ev = None
                else:
                    # everything the caller's stream holds now is ordered before the side stream's work: a depth /
                    # input_pts tensor a GPU op on that stream is still producing, the TSDF update of a synchronous
                    # integrate().  (It also holds the previous frame's decode, so this frame's encode no longer
                    # overlaps it: callers whose frames are complete in device memory set inputs_resident.)
                    enc.wait_stream(main)
                with torch.cuda.stream(enc):
                    feats, pcounts, flat_ids, grid_ids, counters, cap = self._encode_frame_async(frame)
                    done = torch.cuda.Event()
                    done.record(enc)
                    # the TSDF side fusion depends on the frame only as well (gated on the device by the encode's
                    # point count): behind the encode on this stream it runs beside this frame's upsert / decode
                    # instead of between its encoder and decoder
                    self._integrate_tsdf(frame, gate=counters[0:1])
                    tsdf_ev = torch.cuda.Event()
                    tsdf_ev.record(enc)
                main.wait_event(done)
                for t in (feats, pcounts, flat_ids, grid_ids, counters):
                    t.record_stream(main)     # allocated on the encode stream, consumed on the main stream
            else:
                feats, pcounts, flat_ids, grid_ids, counters, cap = self._encode_frame_async(frame)
                self._integrate_tsdf(frame, gate=counters[0:1])
            n_dev = counters[2:3]
            host = torch.empty(16, dtype=torch.int32, pin_memory=True)
            # (the upsert also stamps the decode's origins: k_lattice_stamp's launch goes)
            v.integrate(grid_ids, feats, pcounts, n_dev=n_dev, stamp_origins=decode)
            sdf = v.decode_lattice(grid_ids, self.pointnet.nerf, self.sdf_delta, query_tensor=False,
                                   n_dev=n_dev, prestamped=True) if decode else None
            # counters + {row count, sticky error} in one small launch that writes the pinned words directly
            _lib.check(v._lib.bnv_readback_words(_lib.ptr(counters), _lib.ptr(v._status), C.c_void_p(host.data_ptr()),
                                                 _lib.stream_ptr()), "bnv_readback_words")
            host_rows = host[8:10]
            if self.overlap_encode:
                main.wait_event(tsdf_ev)      # long finished by now: the frame's event covers its TSDF update too
            ev = torch.cuda.Event()
            ev.record()
        return FrameHandle(self, (feats, pcounts, flat_ids, grid_ids), host, ev, cap, sdf, host_rows)

    def optimize(self, n_iters, last_frame=-1, sampling_size=5000, train_ray_splits=1000, ray_max_dist=3,
                 lr=0.001, generator=None):
        """run_e2e.py:111-162: ``n_iters`` Adam steps on the volume features; every step samples
        ``sampling_size`` rays of one random key frame of ``self.frames[last_frame:]`` (dicts with ``depth``
        [H, W] uint16 mm / float m, ``intr_mat``, ``T_wc`` on the device -- the reference re-reads the depth
        png in DataLoader workers, fusion_inference_dataset.py:329-420).  Defaults are the values of
        fusion_pointnet_model.yaml / fusion_inference_dataset.yaml."""
        from .optimize import optimize_volume, sample_key_frame
        self._drain_pipe()
        delta = self.prepare_tsdf_volume() if self.tsdf_vol is not None else self.sdf_delta
        lo = 0 if last_frame == -1 else last_frame
        cpu_gen = generator if (generator is not None and generator.device.type == "cpu") else None

        from .optimize import key_frame_points
        # per key frame: every pixel's world point + validity (4.9 MB per 640x480 frame; the 256 used last are kept,
        # across calls: the reference's DataLoader workers re-read the depth image beside the optimiser)
        cache = self.__dict__.setdefault("_key_frame_points", {})

        def batches():
            for _ in range(n_iters):
                k = int(torch.randint(lo, len(self.frames), (1,), generator=cpu_gen))
                f = self.frames[k]
                hit = cache.get(id(f))
                pts = hit[1] if hit is not None and hit[0] is f and hit[2] == ray_max_dist else None
                if pts is None:
                    d = f["depth"]
                    if d.dtype in (torch.uint16, torch.int16):
                        d = d.to(torch.float32) / 1000.0
                    pts = key_frame_points(d, f["intr_mat"], f["T_wc"], ray_max_dist)
                    if len(cache) >= 256:
                        cache.pop(next(iter(cache)))
                    cache[id(f)] = (f, pts, ray_max_dist)
                yield sample_key_frame(None, None, None, sampling_size, ray_max_dist, generator, points=pts)

        return optimize_volume(self.volume, self.pointnet.nerf, batches(), self.truncated_units,
                               self.truncated_dist, ray_max_dist, sdf_delta=delta,
                               train_ray_splits=train_ray_splits, lr=lr, generator=generator)

    def extract_sdf(self):
        """run_e2e.py:164-167 up to (not including) marching cubes."""
        self._drain_pipe()
        self.volume.to_tensor()
        delta = self.prepare_tsdf_volume() if self.tsdf_vol is not None else self.sdf_delta
        return self.volume.meshlize_sdf(self.pointnet.nerf, delta)

    def extract_mesh(self, path=None):
        """run_e2e.py:164-167: mesh of the whole volume (TSDF prior included when enabled) -> TriMesh or None."""
        self._drain_pipe()
        delta = self.prepare_tsdf_volume() if self.tsdf_vol is not None else self.sdf_delta
        self.volume.to_tensor()
        out = self.volume.meshlize(self.pointnet.nerf, delta, path)
        return None if out is None else out[1]

    def save(self, working_dir, scan_id="scan"):
        """run_e2e.py:188-194: the TSDF volume as <scan_id>.npy (metres) and the feature volume as
        final_sparse_volume.pth (sparse_volume.py:835-860)."""
        import os
        import numpy as np
        self._drain_pipe()
        if self.tsdf_vol is not None:
            tsdf, _ = self.tsdf_vol.get_volume()
            np.save(os.path.join(working_dir, scan_id + ".npy"), tsdf * (self.tsdf_voxel_size * 5))
        self.volume.to_tensor()
        self.volume.save(os.path.join(working_dir, "final"))
