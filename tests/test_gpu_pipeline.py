"""The per-frame chain behind the C object of csrc/pipeline.hip (FramePipe) and the kernels folded for it
(bnv_volume_integrate_frame, bnv_decode_lattice_stamped, bnv_shard_install_reset); the volume workspace's look-back
words; the volume without its dense row index.  Everything is compared bit for bit (torch.equal) with the per-stage
path the other GPU tests pin against the reference's goldens.  Needs a real MI355X: run with  -m gpu."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", params=["split_f16", "fp32_exact"])
def bnv(request):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (no CPU fallback exists)")
    import bnv_fusion_amd
    bnv_fusion_amd.set_mlp_mode(1 if request.param == "split_f16" else 0)
    yield bnv_fusion_amd
    bnv_fusion_amd.set_mlp_mode(1)


def _frames(n, hw=(240, 320), start=0):
    from bnv_fusion_amd import synthetic
    return [{"depth": torch.from_numpy(synthetic.depth_u16(t, *hw)).to(DEV), "intr_mat": synthetic.intrinsics(*hw),
             "T_wc": synthetic.pose(t)} for t in range(start, start + n)]


def _run_pipe(pipe, frames, depth_in_flight, decode=True):
    """Drives a FramePipe (world 1) with ``depth_in_flight`` frames enqueued ahead of the oldest uncollected one."""
    outs, pend = [], []

    def collect():
        s = pend.pop(0)
        outs.append(pipe.outputs(s, pipe.result(s), copy=True))

    for fr in frames:
        while len(pend) >= min(depth_in_flight, pipe.n_slots):
            collect()
        s = pipe.begin(fr)
        assert pipe.bound(s) == 0
        assert pipe.upsert(s, decode=decode) is None
        pipe.finish(s)
        pend.append(s)
    while pend:
        collect()
    return outs


@pytest.mark.parametrize("streams", [4, 2])
@pytest.mark.parametrize("in_flight", [1, 3])
@pytest.mark.parametrize("kind", ["depth", "points"])
def test_frame_pipe_equals_per_stage_path(bnv, in_flight, kind, streams):
    """One GPU: the pipe's outputs, volume and TSDF volume equal NeuralMap.fuse_and_decode's, frame by frame, with
    one or several frames in flight (slots reused: 14 frames through 4 slots), from depth images (front end fused,
    TSDF side fusion) and from input_pts; an empty frame in the middle.  ``streams``: the four-stream schedule (front
    end / encoder / main chain / blend) and round 3's two-stream one (encode / main)."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.frontend import depth_to_input_pts
    from bnv_fusion_amd.pipeline import FramePipe
    from bnv_fusion_amd.sparse_volume import get_world_range
    from bnv_fusion_amd.tsdf import TSDFVolume
    dims, voxel = synthetic.GRID_DIMS[128]             # 0.02 m voxels: weights reach min_pts within 8 frames
    dims3 = np.array([dims] * 3)
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    frames = _frames(14)
    far = dict(frames[5])
    T = np.array(far["T_wc"], dtype=np.float64)
    T[:3, 3] += 50.0                                   # no point inside the volume
    far["T_wc"] = T
    frames[5] = far
    if kind == "points":
        frames = [{"input_pts": depth_to_input_pts(f["depth"], f["intr_mat"], f["T_wc"], max_depth=3.0,
                                                   compact=False)[0]} for f in frames]
    tsdf = kind == "depth"
    ref_nm = bnv.NeuralMap(dims3, voxel, model, device=DEV, tsdf=tsdf)
    ref = [ref_nm.fuse_and_decode(f) for f in frames]
    vol = bnv.SparseVolume(8, voxel, dims3, 8, device=DEV)
    tv = None
    if tsdf:
        mn, mx, _ = get_world_range(dims3, 0.025)
        tv = TSDFVolume(np.stack([mn, mx], 1), 0.025, device=DEV)
    pipe = FramePipe(vol, model, 240 * 320, n_slots=4, tsdf_vol=tv, streams=streams)
    assert pipe.double_buffered == (streams == 4)
    got = _run_pipe(pipe, frames, in_flight)
    torch.cuda.synchronize()
    for t, ((rc, rs), (gc, gs)) in enumerate(zip(ref, got)):
        if rc is None:
            assert gc is None and gs is None and t == 5
            continue
        assert torch.equal(rc, gc), t
        assert torch.equal(rs, gs), t
    assert float((ref[-1][1] != voxel).float().mean()) > 0.05           # the decode mask is live
    a, b = ref_nm.volume, vol
    n = a.num_rows()
    assert b.num_rows() == n
    assert torch.equal(a._row_coords[:n], b._row_coords[:n]) and torch.equal(a._features[:n], b._features[:n])
    assert torch.equal(a._weights[:n], b._weights[:n])
    assert vol.n_frames == ref_nm.volume.n_frames and vol.n_pts_list == ref_nm.volume.n_pts_list
    if tsdf:
        assert torch.equal(tv.tsdf, ref_nm.tsdf_vol.tsdf) and torch.equal(tv.weight, ref_nm.tsdf_vol.weight)
    pipe.close()


def test_frame_pipe_grows_the_volume_mid_stream(bnv):
    """A volume of the reference's initial capacity grows (re-allocation + re-hash, new lattice workspace) while
    frames are in flight: same outputs as the per-stage path."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.pipeline import FramePipe
    dims, voxel = synthetic.GRID_DIMS[128]
    dims3 = np.array([dims] * 3)
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    frames = _frames(12, hw=(480, 640))
    ref_nm = bnv.NeuralMap(dims3, voxel, model, device=DEV, capacity=1 << 20)
    ref = [ref_nm.fuse_and_decode(f) for f in frames]
    vol = bnv.SparseVolume(8, voxel, dims3, 8, capacity=2000, device=DEV)
    cap0 = vol._row_capacity
    got = _run_pipe(FramePipe(vol, model, 480 * 640, n_slots=3), frames, 2)
    assert vol._row_capacity > cap0
    for (rc, rs), (gc, gs) in zip(ref, got):
        assert torch.equal(rc, gc) and torch.equal(rs, gs)
    assert float((ref[-1][1] != voxel).float().mean()) > 0.05


def test_integrate_frame_packs_the_records_bnv_shard_pack_does(bnv):
    """bnv_volume_integrate_frame on a sharded grid: the records it appends are the ones bnv_shard_pack lists for the
    same keys after a plain upsert (as sets; record order is free), rows / values equal the plain upsert's."""
    import ctypes as C
    from bnv_fusion_amd import _lib, distributed as D
    from bnv_fusion_amd.sparse_volume import make_grid
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    dims, voxel = z["dims"], float(z["voxel_size"])
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    lib = _lib.load()
    world, rank = 3, 1
    vols = [bnv.SparseVolume(8, voxel, dims, 8, device=DEV) for _ in range(2)]
    for v in vols:
        v.shard = (rank, world, D.BLOCK_LOG2)
        v._grid = make_grid(v._n_xyz_host, v.min_coords, v.max_coords, voxel, 8, v.shard)
    for k, fr in enumerate(z["frames"][:6]):
        f, c, ids, g, n = model.encode_pointcloud(torch.from_numpy(fr).to(DEV), vols[0].n_xyz, vols[0].min_coords,
                                                  vols[0].max_coords, voxel, return_dense=False)
        own = torch.from_numpy(D.voxel_owner(g.cpu().numpy(), world) == rank).to(DEV)
        g, f, c = g[own].contiguous(), f[own].contiguous(), c[own, 0].contiguous()
        m = int(g.shape[0])
        a, b = vols
        a.integrate(g, f, c)
        blk_a = torch.zeros((m + 1) * D.REC_WORDS, dtype=torch.int32, device=DEV)
        _lib.check(lib.bnv_shard_pack(C.byref(a._struct()), C.byref(a._grid), _lib.ptr(g), m, None, _lib.ptr(blk_a), m,
                                      _lib.stream_ptr()), "bnv_shard_pack")
        b._reserve(m)
        ws = b._workspace(m)
        lws, ep = b._lattice_workspace(m)
        blk_b = torch.zeros((m + 1) * D.REC_WORDS, dtype=torch.int32, device=DEV)
        blk_b[1] = rank
        x = _lib.IntegrateExtras()
        x.shard_block, x.shard_block_capacity, x.grid_host = blk_b.data_ptr(), m, C.pointer(b._grid)
        x.lattice_ws, x.stamp_epoch = lws.data_ptr(), ep
        _lib.check(lib.bnv_volume_integrate_frame(C.byref(b._struct()), _lib.ptr(g), _lib.ptr(f), _lib.ptr(c), m, None,
                                                  _lib.ptr(ws), ws.numel(), C.byref(x), _lib.stream_ptr()),
                   "bnv_volume_integrate_frame")
        b._rows_upper += m
        ra = blk_a.view(m + 1, D.REC_WORDS).cpu().numpy()
        rb = blk_b.view(m + 1, D.REC_WORDS).cpu().numpy()
        na, nb = int(ra[0, 0]), int(rb[0, 0])
        assert na == nb > 0 and list(rb[0, :3]) == [nb, rank, 0]
        ka, kb = ra[1: 1 + na], rb[1: 1 + nb]
        assert np.array_equal(ka[np.lexsort(ka[:, :3].T[::-1])], kb[np.lexsort(kb[:, :3].T[::-1])]), k
        n_rows = a.num_rows()
        assert b.num_rows() == n_rows and torch.equal(a._row_coords[:n_rows], b._row_coords[:n_rows])
        assert torch.equal(a._features[:n_rows], b._features[:n_rows]) and torch.equal(a._weights[:n_rows], b._weights[:n_rows])
        # the stamped decode equals the plain one
        sa = a.decode_lattice(g, model.nerf, query_tensor=False)
        sb = torch.empty((m, 27), dtype=torch.float32, device=DEV)
        d, _ = b._delta(None)
        _lib.check(lib.bnv_decode_lattice_stamped(C.byref(b._struct()), C.byref(b._grid), _lib.ptr(b._features),
                                                  _lib.ptr(b._weights), b._row_capacity, _lib.ptr(model.nerf.sdf_pack),
                                                  _lib.ptr(g), m, None, C.byref(d), _lib.ptr(lws), lws.numel(), ep,
                                                  _lib.ptr(sb), _lib.stream_ptr()), "bnv_decode_lattice_stamped")
        assert torch.equal(sa, sb), k


def test_volume_workspace_lookback_words_never_alias(bnv):
    """the look-back words of the volume workspace used to sit behind arrays sized by the call's n, so a
    small call's words landed in the slot indices a larger call had left there.  Large and small insert / integrate /
    integrate_batch calls alternate on ONE volume (one workspace, stale words and all); its rows must equal those of
    a volume that runs the same calls with a zero-filled workspace each, and be the keys in first-occurrence order."""
    rng = np.random.default_rng(3)
    dims = np.array([2.54] * 3)

    def calls():
        # (kind, keys): sizes chosen so that small calls' tile words would fall inside large calls' slot_of arrays
        out = []
        nxt = 0
        for rep in range(40):
            for kind, n in (("batch", 70000), ("integrate", 300), ("insert", 50000), ("batch", 900), ("integrate", 5000)):
                fresh = np.arange(nxt, nxt + n // 2)
                nxt += n // 2
                old = rng.integers(0, max(nxt - n // 2, 1), n - n // 2)
                ids = np.unique(np.concatenate([fresh, old]))
                rng.shuffle(ids)
                out.append((kind, ids))
        return out

    def keys_of(ids):
        return torch.from_numpy(np.stack([ids // 65536, (ids // 256) % 256, ids % 256], 1)).to(DEV)

    def apply(vol, kind, ids, fresh_ws):
        k = keys_of(ids)
        n = len(ids)
        f = torch.full((n, 8), 0.5, device=DEV)
        c = torch.full((n,), 16, dtype=torch.int64, device=DEV)
        if fresh_ws:
            vol._ws = None
        if kind == "integrate":
            vol.integrate(k, f, c)
        elif kind == "insert":
            vol.insert(k, f, torch.ones(n, device=DEV), torch.zeros(n, device=DEV))
        else:
            h = n // 3
            vol.integrate_batch([(k[:h], f[:h], c[:h], None), (k[h:], f[h:], c[h:], None)])

    seq = calls()
    a = bnv.SparseVolume(8, 0.01, dims, 8, capacity=1 << 22, device=DEV)
    b = bnv.SparseVolume(8, 0.01, dims, 8, capacity=1 << 22, device=DEV)
    for kind, ids in seq:
        apply(a, kind, ids, fresh_ws=False)      # one shared workspace, stale words and all
        apply(b, kind, ids, fresh_ws=True)       # a zero-filled workspace per call
    n = a.num_rows()
    assert n == b.num_rows() and n > 1_000_000
    assert torch.equal(a._row_coords[:n], b._row_coords[:n])
    assert torch.equal(a._weights[:n], b._weights[:n])
    # and the rows are the keys in first-occurrence order (a batch numbers its frames in order: the same order)
    seen = np.zeros(max(int(ids.max()) for _, ids in seq) + 1, dtype=bool)
    order = []
    for _, ids in seq:
        new = ids[~seen[ids]]
        seen[new] = True
        order.append(new)
    rc = a._row_coords[:n].cpu().numpy()
    assert np.array_equal(rc[:, 0] * 65536 + rc[:, 1] * 256 + rc[:, 2], np.concatenate(order))


def test_volume_without_dense_row_index(bnv):
    """``brick=False``: the hash alone serves every look-up; decode and upserts equal the indexed volume's."""
    from bnv_fusion_amd import synthetic
    dims, voxel = synthetic.GRID_DIMS[128]
    dims3 = np.array([dims] * 3)
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    frames = _frames(10)
    vols = [bnv.SparseVolume(8, voxel, dims3, 8, device=DEV, brick=flag) for flag in (None, False)]
    assert vols[0]._brick is not None and vols[1]._brick is None
    for fr in frames:
        outs = []
        for v in vols:
            f, c, fl, g, cnt, cap = model.encode_depth_async(fr["depth"], fr["intr_mat"], fr["T_wc"], 3.0, v.n_xyz,
                                                             v.min_coords, v.max_coords, voxel)[:6]
            n = int(cnt[2])
            v.integrate(g[:n], f[:n], c[:n])
            outs.append(v.decode_lattice(g[:n], model.nerf, query_tensor=False))
        assert torch.equal(outs[0], outs[1])
    assert float((outs[0] != voxel).float().mean()) > 0.05
    n = vols[0].num_rows()
    assert vols[1].num_rows() == n and torch.equal(vols[0]._features[:n], vols[1]._features[:n])


def test_side_streams_are_verified_to_overlap():
    """streams.concurrent_stream hands out a stream that a spin-kernel test has shown to run beside the caller's; two
    streams forced onto one hardware queue are told apart by the same test."""
    import ctypes as C
    from bnv_fusion_amd import _lib, streams
    lib = _lib.require_device(0)
    main = torch.cuda.current_stream()
    s = streams.concurrent_stream(DEV, main)
    assert s.cuda_stream != main.cuda_stream and s.bnv_concurrent
    ok, one, two = streams._overlaps(lib, main, s)
    assert ok and two < 1.6 * one
    same, one2, two2 = streams._overlaps(lib, main, main)          # the same stream: strictly one after the other
    assert not same and two2 > 1.8 * one2
    assert lib.bnv_probe_spin(0, 10, None) != 0 and lib.bnv_probe_spin(1, -1, None) != 0      # bad arguments are refused
    t = streams.concurrent_stream(DEV, main, exclude=(s,))
    assert t.cuda_stream not in (main.cuda_stream, s.cuda_stream)
    # the pipes of a process share ONE verified set per (device, main stream): a second pipe does not draw new candidates
    a = streams.pipe_streams(DEV, main, 3)
    b = streams.pipe_streams(DEV, main, 3)
    assert len(a) == 3 and all(x is y for x, y in zip(a, b)) and all(x.bnv_concurrent for x in a)
    assert len({x.cuda_stream for x in a} | {main.cuda_stream}) == 4
    assert streams.pipe_streams(DEV, main, 1)[0] is a[0]


def test_frame_timeline_is_ordered_and_changes_nothing(bnv):
    """bnv_frame_pipe_timeline_enable / bnv_frame_timeline: with the diagnostic on, every stage of a frame leaves a GPU
    timestamp, in stage order on each stream; the frames' results are those of a pipe without it."""
    import ctypes as C
    from bnv_fusion_amd import _lib, synthetic
    from bnv_fusion_amd.pipeline import FramePipe
    dims, voxel = synthetic.GRID_DIMS[128]
    dims3 = np.array([dims] * 3)
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    frames = _frames(10)
    outs = []
    for on in (False, True):
        vol = bnv.SparseVolume(8, voxel, dims3, 8, device=DEV)
        pipe = FramePipe(vol, model, 240 * 320, n_slots=4)
        lib = pipe._lib
        assert lib.bnv_frame_timeline(pipe._h, 0, (C.c_float * 11)()) != 0          # never enabled
        if on:
            _lib.check(lib.bnv_frame_pipe_timeline_enable(pipe._h, 1), "enable")
        res = []
        for fr in frames:
            s = pipe.begin(fr)
            pipe.upsert(s)
            pipe.finish(s)
            if on:
                assert lib.bnv_frame_timeline(pipe._h, s, (C.c_float * 11)()) != 0  # the frame has not been collected
            c, sdf = pipe.outputs(s, pipe.result(s), copy=True)
            res.append((c, sdf))
            if on:
                t = (C.c_float * 11)()
                _lib.check(lib.bnv_frame_timeline(pipe._h, s, t), "bnv_frame_timeline")
                t = np.array(t[:])
                assert np.isfinite(t).all(), t
                for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10)):
                    assert t[a] <= t[b], (a, b, t)      # (one frame at a time: the stages do not overlap other frames)
                assert t[10] - t[0] < 50.0                                           # ms
        outs.append(res)
        pipe.close()
    for (c0, s0), (c1, s1) in zip(*outs):
        assert torch.equal(c0, c1) and torch.equal(s0, s1)


def test_frame_pipe_integrate_only_frames_and_tsdf_prior(bnv):
    """FramePipe with frames that are only fused (decode=False: run_e2e.py's integrate) followed by frames decoded WITH
    the TSDF prior (sdf_delta, sparse_volume.py:819-832) and a smaller frame in between: equal to NeuralMap."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.pipeline import FramePipe
    dims, voxel = synthetic.GRID_DIMS[128]
    dims3 = np.array([dims] * 3)
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    frames = _frames(12, hw=(240, 320))
    small = _frames(1, hw=(120, 160), start=5)[0]
    ref_nm = bnv.NeuralMap(dims3, voxel, model, device=DEV, tsdf=True)
    vol = bnv.SparseVolume(8, voxel, dims3, 8, device=DEV)
    pipe = FramePipe(vol, model, 240 * 320, n_slots=3, tsdf_vol=None)
    for fr in frames[:8]:                                  # fuse only
        ref_c = ref_nm.integrate(fr)
        s = pipe.begin(fr)
        pipe.bound(s)
        pipe.upsert(s, decode=False)
        pipe.finish(s)
        c, sdf = pipe.outputs(s, pipe.result(s))
        assert sdf is None and torch.equal(c, ref_c)
    delta = ref_nm.prepare_tsdf_volume()                   # [1, 1, X, Y, Z] prior from the 8 fused frames
    ref_nm.sdf_delta = delta
    pipe.sdf_delta = delta
    for fr in frames[8:10] + [small] + frames[10:]:
        rc, rs = ref_nm.fuse_and_decode(fr)
        s = pipe.begin(fr)
        pipe.bound(s)
        pipe.upsert(s, decode=True)
        pipe.finish(s)
        c, sdf = pipe.outputs(s, pipe.result(s))
        assert torch.equal(c, rc) and torch.equal(sdf, rs)
    assert float((rs != voxel).float().mean()) > 0.05
    plain = ref_nm.volume.decode_lattice(rc, model.nerf, None, query_tensor=False)
    assert not torch.equal(plain, rs)                      # the prior does change the decode


@pytest.mark.parametrize("with_rgb", [False, True])
def test_points_frames_with_depth_update_the_tsdf_volume(bnv, with_rgb):
    """The reference dataset's frames carry input_pts, the depth image and the colour image together
    (run_e2e.py:78-109): through the frame pipeline the TSDF side volume must receive the depth exactly as through the
    per-stage path (NeuralMap.frame_pipe = False) -- round 4's pipe dropped it for frames that hold input_pts."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.frontend import depth_to_input_pts
    dims, voxel = synthetic.GRID_DIMS[128]
    dims3 = np.array([dims] * 3)
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    g = torch.Generator().manual_seed(2)
    frames = []
    for k, f in enumerate(_frames(10)):
        f = dict(f)
        f["input_pts"] = depth_to_input_pts(f["depth"], f["intr_mat"], f["T_wc"], max_depth=3.0, compact=False)[0]
        if k == 4:                                      # metres as float32 instead of uint16 millimetres
            f["depth"] = f["depth"].to(torch.float32) / 1000.0
        if with_rgb:
            f["rgb"] = (torch.rand(240, 320, 3, generator=g) * 255).floor().to(DEV)
        frames.append(f)
    far = dict(frames[6])
    far["input_pts"] = far["input_pts"] + 50.0          # no point inside: the reference returns before the TSDF fusion
    frames[6] = far
    maps = []
    for use_pipe in (False, True):
        nm = bnv.NeuralMap(dims3, voxel, model, device=DEV, tsdf=True)
        nm.frame_pipe = use_pipe
        hs = [nm.fuse_and_decode_async(f) for f in frames[:3]]
        outs = [h.result() for h in hs]
        for f in frames[3:]:
            outs.append(nm.fuse_and_decode_async(f).result())
        torch.cuda.synchronize()
        maps.append((nm, outs))
    (a, oa), (b, ob) = maps
    assert b._pipe is not None and a._pipe is None
    for (ca, sa), (cb, sb) in zip(oa, ob):
        assert (ca is None) == (cb is None)
        if ca is not None:
            assert torch.equal(ca, cb) and torch.equal(sa, sb)
    assert oa[6] == (None, None)
    assert float((a.tsdf_vol.weight > 0).float().mean()) > 0.01         # the side volume did receive the frames
    assert torch.equal(a.tsdf_vol.tsdf, b.tsdf_vol.tsdf) and torch.equal(a.tsdf_vol.weight, b.tsdf_vol.weight)
    if with_rgb:
        assert torch.equal(a.tsdf_vol.color, b.tsdf_vol.color) and float(a.tsdf_vol.color.abs().sum()) > 0
    # nine frames fused, not ten: the frame without a point inside the volume left the TSDF volume alone
    assert float(a.tsdf_vol.weight.max()) == 9.0


def test_frame_cancel_frees_the_slot_and_leaves_the_volume_alone(bnv):
    """bnv_frame_cancel: a frame that was begun (its encode is enqueued) and is never upserted gives its slot back;
    the frames around it come out as if it had never been begun."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.pipeline import FramePipe
    dims, voxel = synthetic.GRID_DIMS[128]
    dims3 = np.array([dims] * 3)
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    frames = _frames(11)
    ref_nm = bnv.NeuralMap(dims3, voxel, model, device=DEV)
    ref = [ref_nm.fuse_and_decode(f) for f in frames[:5] + frames[6:]]
    vol = bnv.SparseVolume(8, voxel, dims3, 8, device=DEV)
    pipe = FramePipe(vol, model, 240 * 320, n_slots=2)
    got = []
    for t, fr in enumerate(frames):
        s = pipe.begin(fr)
        if t == 5:
            with pytest.raises(Exception):
                pipe.begin(frames[0], slot=s)          # the slot is busy
            pipe.cancel(s)
            with pytest.raises(Exception):
                pipe.cancel(s)                         # nothing begun in it any more
            continue
        pipe.bound(s)
        pipe.upsert(s)
        pipe.finish(s)
        got.append(pipe.outputs(s, pipe.result(s)))
    for (rc, rs), (gc, gs) in zip(ref, got):
        assert torch.equal(rc, gc) and torch.equal(rs, gs)
    assert vol.num_rows() == ref_nm.volume.num_rows()


def test_slot_reuse_ahead_of_the_upsert_is_not_an_error(bnv):
    """Three slots, the NEXT frame begun before this frame's upsert (ShardedNeuralMap(next_frame=...)): slot (t + 1) is
    the slot frame t - 2 used, and it is begun again (state 1) before upsert(t) looks up the decode workspace that
    frame t - 2 used last.  Round 4's hazard check took that for an unfinished frame and refused the upsert."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.pipeline import FramePipe
    dims, voxel = synthetic.GRID_DIMS[128]
    dims3 = np.array([dims] * 3)
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    frames = _frames(12)
    ref_nm = bnv.NeuralMap(dims3, voxel, model, device=DEV)
    ref = [ref_nm.fuse_and_decode(f) for f in frames]
    vol = bnv.SparseVolume(8, voxel, dims3, 8, capacity=3000, device=DEV)      # it also grows on the way
    pipe = FramePipe(vol, model, 240 * 320, n_slots=3)
    got, pend = [], []
    pre = pipe.begin(frames[0])
    for t in range(len(frames)):
        s = pre
        while len(pend) >= 1:                          # one finished frame uncollected + this one + the next = 3 slots
            q = pend.pop(0)
            got.append(pipe.outputs(q, pipe.result(q)))
        pre = pipe.begin(frames[t + 1]) if t + 1 < len(frames) else None
        pipe.bound(s)
        pipe.upsert(s)
        pipe.finish(s)
        pend.append(s)
    while pend:
        q = pend.pop(0)
        got.append(pipe.outputs(q, pipe.result(q)))
    for (rc, rs), (gc, gs) in zip(ref, got):
        assert torch.equal(rc, gc) and torch.equal(rs, gs)


def test_persistent_tables_carry_entries_over(bnv, monkeypatch):
    """The frame pipeline keeps ONE SDF table per volume across frames (bnv_volume_t.lattice_table / lattice_have): an
    entry in a row the frame did not update is not evaluated again.  Same outputs bit for bit as the pipe without it
    (and as the per-stage path), fewer MLP evaluations; the books stay right through a frame that is only fused, a
    volume growth, an insert behind the pipe's back and a change of the arithmetic mode."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.pipeline import FramePipe, W_EVALS
    dims, voxel = synthetic.GRID_DIMS[128]
    dims3 = np.array([dims] * 3)
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    frames = _frames(26)
    runs = {}
    for on in ("1", "0"):
        monkeypatch.setenv("BNV_PERSISTENT_TABLES", on)
        vol = bnv.SparseVolume(8, voxel, dims3, 8, capacity=30000, device=DEV)      # grows once on the way
        pipe = FramePipe(vol, model, 240 * 320, n_slots=3)
        assert pipe.persistent_tables == (on == "1")
        outs, evals, pend = [], [], []

        def collect():
            s = pend.pop(0)
            w = pipe.result(s)
            evals.append(int(w[W_EVALS]))
            outs.append(pipe.outputs(s, w, copy=True))

        mode0 = bnv.get_mlp_mode()
        for t, fr in enumerate(frames):
            while len(pend) >= 2:
                collect()
            if t == 12:           # features written through the class's own insert: the rows' entries are dropped
                while pend:
                    collect()
                vol.to_tensor()
                k = vol.active_coordinates[::7]
                f, w_, h = vol.query(k)
                vol.insert(k, f * 1.01, w_, h)
            if t == 18:           # another arithmetic for the frames from here on
                model.set_mlp_mode(0 if mode0 == 1 else 1)
            s = pipe.begin(fr)
            pipe.bound(s)
            pipe.upsert(s, decode=t != 9)          # frame 9 is only fused
            pipe.finish(s)
            pend.append(s)
        while pend:
            collect()
        model.set_mlp_mode(None)
        runs[on] = (outs, evals, vol)
        pipe.close()
    (oa, ea, va), (ob, eb, vb) = runs["1"], runs["0"]
    for t, ((ca, sa), (cb, sb)) in enumerate(zip(oa, ob)):
        assert torch.equal(ca, cb), t
        assert (sa is None) == (sb is None) and (sa is None or torch.equal(sa, sb)), t
    assert float((oa[-1][1] != voxel).float().mean()) > 0.05
    assert all(a <= b for a, b in zip(ea, eb))
    late = slice(20, None)      # (a short pan: nearly every row a frame reads is one it has just updated -- the benchmark's
    assert sum(ea[late]) < sum(eb[late]), (sum(ea[late]), sum(eb[late]))     # steady state carries ~5 % over, bench.py)
    n = va.num_rows()
    assert n == vb.num_rows() and torch.equal(va._features[:n], vb._features[:n])


def test_persistent_tables_do_not_survive_a_change_of_model(bnv):
    """A NeuralMap whose networks are swapped mid-stream makes a new frame pipeline on the same volume: table entries the
    old networks computed must not be carried over.  Same outputs as the per-stage path, bit for bit."""
    from bnv_fusion_amd import synthetic
    dims, voxel = synthetic.GRID_DIMS[128]
    dims3 = np.array([dims] * 3)
    m1 = bnv.load_pretrained(device=DEV, voxel_size=voxel).set_mlp_mode(1)
    m0 = bnv.load_pretrained(device=DEV, voxel_size=voxel).set_mlp_mode(0)
    frames = _frames(16)
    outs = {}
    for use_pipe in (True, False):
        nm = bnv.NeuralMap(dims3, voxel, m1, device=DEV)
        nm.frame_pipe = use_pipe
        res = []
        for t, fr in enumerate(frames):
            if t == 11:
                nm.pointnet = m0
            res.append(nm.fuse_and_decode_async(fr).result())
        outs[use_pipe] = res
        if use_pipe:
            assert nm._pipe is not None and nm._pipe.persistent_tables and nm._pipe.pointnet is m0
    for t, ((ca, sa), (cb, sb)) in enumerate(zip(outs[True], outs[False])):
        assert torch.equal(ca, cb) and torch.equal(sa, sb), t
    assert float((outs[True][-1][1] != voxel).float().mean()) > 0.05


def test_persistent_tables_do_not_survive_new_weights_in_the_same_model(bnv):
    """``load_state_dict`` on the SAME model object mid-stream (repack() rewrites nerf.sdf_pack in place, NeuralMap keeps
    its pipe): the carried-over table entries were computed with the old SDF weights and must be dropped -- the pipe
    watches the network's pack version.  Same outputs as the per-stage path, bit for bit; and they DO differ from a run
    that keeps the old weights (the change is visible in the decode)."""
    from bnv_fusion_amd import synthetic
    dims, voxel = synthetic.GRID_DIMS[128]
    dims3 = np.array([dims] * 3)
    frames = _frames(16)
    outs = {}
    for use_pipe in (True, False):
        m = bnv.load_pretrained(device=DEV, voxel_size=voxel)
        nm = bnv.NeuralMap(dims3, voxel, m, device=DEV)
        nm.frame_pipe = use_pipe
        res, pipe0 = [], None
        for t, fr in enumerate(frames):
            if t == 11:
                pipe0 = nm._pipe
                v0 = m.nerf.pack_version
                sd = {k: v.clone() for k, v in m.state_dict().items()}
                sd["nerf.fc_alpha.bias"] += 0.25
                sd["nerf.geo_layer1.weight"] *= 1.02
                m.load_state_dict(sd)
                assert m.nerf.pack_version == v0 + 1
            res.append(nm.fuse_and_decode_async(fr).result())
        outs[use_pipe] = res
        if use_pipe:
            assert nm._pipe is pipe0 and pipe0 is not None and pipe0.persistent_tables      # the pipe was kept
    for t, ((ca, sa), (cb, sb)) in enumerate(zip(outs[True], outs[False])):
        assert torch.equal(ca, cb) and torch.equal(sa, sb), t
    live = outs[True][-1][1] != voxel
    assert float(live.float().mean()) > 0.05
    # the same frames with the old weights throughout: the last frames decode differently
    m = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    nm = bnv.NeuralMap(dims3, voxel, m, device=DEV)
    old = [nm.fuse_and_decode_async(fr).result() for fr in frames]
    assert torch.equal(old[10][1], outs[True][10][1]) and not torch.equal(old[-1][1], outs[True][-1][1])
