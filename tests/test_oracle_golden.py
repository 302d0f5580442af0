"""Pins oracle/bnv_oracle.py against the vectors captured from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import bnv_oracle as orc
from conftest import GOLDEN, WEIGHTS_FP32

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
FLOAT_TOL = 2e-6  # same ATen ops as the reference; only thread-count summation order may differ


@pytest.fixture(scope="module")
def sd():
    return orc.load_weights(WEIGHTS_FP32)


def _vol(z):
    return orc.OracleSparseVolume(8, float(z["voxel_size"]), z["dims"], 8)


@pytest.mark.parametrize("name", ["encode_64", "encode_128"])
def test_encode_sparse_matches_reference(sd, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    vol = _vol(z)
    pts = torch.from_numpy(z["input_pts"])
    f, c, ids, g, n = orc.encode_pointcloud(sd, pts, vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size)
    assert np.array_equal(ids.numpy(), z["flat_ids"])          # bit-exact
    assert np.array_equal(c.numpy(), z["pcounts"]) and c.dtype == torch.int64
    assert np.array_equal(g.numpy(), z["grid_ids"])
    assert float(n) == float(z["n_avg_pts"])
    assert np.abs(f.numpy() - z["feats"]).max() <= FLOAT_TOL


def test_encode_dense_matches_reference(sd):
    z = np.load(os.path.join(GOLDEN, "encode_64.npz"))
    vol = _vol(z)
    fg, mask, uids, flat_all = orc.encode_pointcloud(
        sd, torch.from_numpy(z["input_pts"]), vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size,
        return_dense=True)
    assert np.array_equal(uids.numpy(), z["dense_unique_flat_ids"])
    assert np.array_equal(flat_all[0].numpy(), z["dense_flat_ids_all"].astype(np.int64))
    assert np.array_equal(mask[0, 0].reshape(-1).nonzero()[:, 0].numpy(), z["dense_nonzero"])
    assert np.array_equal(mask[0, 0].reshape(-1)[uids].numpy(), z["dense_counts"])
    assert np.abs(fg[0].reshape(8, -1)[:, uids].T.numpy() - z["dense_feats"]).max() <= FLOAT_TOL


def test_encode_empty_returns_none(sd):
    z = np.load(os.path.join(GOLDEN, "encode_64.npz"))
    vol = _vol(z)
    pts = torch.from_numpy(z["input_pts"]).clone()
    pts[..., :3] += 100.0
    out = orc.encode_pointcloud(sd, pts, vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size)
    assert out == (None,) * 5


@pytest.fixture(scope="module")
def fused_volume(sd):
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    vol = _vol(z)
    for fr in z["frames"]:
        f, c, _, g, n = orc.encode_pointcloud(sd, torch.from_numpy(fr), vol.n_xyz, vol.min_coords,
                                              vol.max_coords, vol.voxel_size)
        vol.track_n_pts(n)
        orc.integrate(vol, g, f, c)
    vol.to_tensor()
    return vol, z


def test_sequence_volume_matches_reference(fused_volume):
    vol, z = fused_volume
    assert np.array_equal(vol.active_coordinates.numpy(), z["keys_insertion"])
    k = vol.active_coordinates.numpy()
    order = np.lexsort((k[:, 2], k[:, 1], k[:, 0]))
    assert np.array_equal(k[order], z["keys_sorted"])
    assert np.abs(vol.weights.numpy()[order] - z["weights_sorted"]).max() <= 1e-6
    assert np.abs(vol.features.numpy()[order] - z["features_sorted"]).max() <= FLOAT_TOL
    assert np.array_equal(vol.num_hits.numpy()[order], z["num_hits_sorted"])
    assert np.allclose(vol.n_pts_list, z["n_pts_list"])


def test_decode_pts_matches_reference(sd, fused_volume):
    vol, _ = fused_volume
    z = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    lat = torch.from_numpy(z["lattice_coords"])
    rnd = torch.from_numpy(z["random_coords"])
    delta = torch.from_numpy(z["sdf_delta"])
    assert np.array_equal(orc.lattice_coords(z["origins"]).numpy(), z["lattice_coords"])
    cases = {
        "lattice_qt": vol.decode_pts(lat, sd, None, is_coords=True, query_tensor=True),
        "lattice_q": vol.decode_pts(lat, sd, None, is_coords=True, query_tensor=False),
        "lattice_delta": vol.decode_pts(lat, sd, delta, is_coords=True, query_tensor=True),
        "random_qt": vol.decode_pts(rnd, sd, None, is_coords=True, query_tensor=True),
        "random_world_out": vol.decode_pts(torch.from_numpy(z["random_world_coords"]), sd, None,
                                           is_coords=False, query_tensor=False),
        "random_delta": vol.decode_pts(rnd, sd, delta, is_coords=True, query_tensor=True),
    }
    for k, v in cases.items():
        assert v.shape == z[k].shape, k
        assert np.abs(v.numpy() - z[k]).max() <= FLOAT_TOL, k
        assert np.array_equal(v.numpy() == vol.voxel_size, z[k] == np.float32(vol.voxel_size)), k
    # count_optim then decode
    w0 = vol.weights.clone()
    vol.count_optim(orc.get_neighbors(rnd))
    k = vol.active_coordinates.numpy()
    order = np.lexsort((k[:, 2], k[:, 1], k[:, 0]))
    assert np.array_equal(vol.weights.numpy()[order], z["weights_after_count_optim_sorted"])
    out = vol.decode_pts(rnd, sd, None, is_coords=True, query_tensor=True)
    assert np.abs(out.numpy() - z["random_after_count_optim"]).max() <= FLOAT_TOL
    vol.weights.copy_(w0)


def test_dense_decode_matches_reference(sd):
    z = np.load(os.path.join(GOLDEN, "dense_decode_64.npz"))
    vol = _vol(z)
    fg, mask, _, _ = orc.encode_pointcloud(sd, torch.from_numpy(z["input_pts"]), vol.n_xyz, vol.min_coords,
                                           vol.max_coords, vol.voxel_size, return_dense=True)
    sdf, _ = orc.decode_feature_grid_w_pts(sd, torch.from_numpy(z["queries"]), fg, mask, vol.voxel_size)
    assert sdf.shape == z["sdf"].shape
    assert np.abs(sdf.numpy() - z["sdf"]).max() <= FLOAT_TOL
    assert 0 < (z["sdf"] == np.float32(vol.voxel_size)).sum() < z["sdf"].size


def test_dense_decode_other_branches_match_reference(sd):
    """global_coords=True (the signature default) and interpolate_decode=False, local_point_fusion.py:288-292, 331-367."""
    z0 = np.load(os.path.join(GOLDEN, "dense_decode_64.npz"))
    z = np.load(os.path.join(GOLDEN, "dense_modes_64.npz"))
    vol = _vol(z0)
    fg, mask, _, _ = orc.encode_pointcloud(sd, torch.from_numpy(z0["input_pts"]), vol.n_xyz, vol.min_coords,
                                           vol.max_coords, vol.voxel_size, return_dense=True)
    q = torch.from_numpy(z["queries"])
    v = np.float32(vol.voxel_size)
    for name, kw in (("global", dict(global_coords=True)), ("nearest", dict(interpolate_decode=False))):
        sdf, nf = orc.decode_feature_grid_w_pts(sd, q, fg, mask, vol.voxel_size, **kw)
        ref = z[f"sdf_{name}"]
        assert sdf.shape == ref.shape and nf.shape == z[f"feats_{name}"].shape
        assert np.array_equal(sdf.numpy() == v, ref == v)
        assert np.abs(sdf.numpy() - ref).max() <= FLOAT_TOL
        assert np.abs(nf.numpy() - z[f"feats_{name}"]).max() <= FLOAT_TOL
        assert 0.05 < (ref == v).mean() < 0.5


def test_known_answers(sd, fused_volume):
    """SURVEY.md section 8c known-answer properties (no reference import needed)."""
    vol, z = fused_volume
    # (iv) flatten(unflatten(id)) == id
    ids = torch.randint(0, 64 ** 3, (1000,))
    assert torch.equal(orc.flatten(orc.unflatten(ids, vol.n_xyz), vol.n_xyz), ids)
    # (iii) a lattice point with a missing corner decodes to exactly voxel_size
    far = orc.lattice_coords(np.array([[3, 3, 3]]))
    assert torch.all(vol.decode_pts(far, sd, None, is_coords=True) == vol.voxel_size)
    # (i) idempotence: fusing one frame k times keeps features, weights = k * min(count/32, 1)
    v2 = _vol(z)
    fr = torch.from_numpy(z["frames"][0])
    f, c, _, g, _ = orc.encode_pointcloud(sd, fr, v2.n_xyz, v2.min_coords, v2.max_coords, v2.voxel_size)
    for _ in range(3):
        orc.integrate(v2, g, f, c)
    v2.to_tensor()
    assert np.abs(v2.features.numpy() - f.numpy()).max() < 5e-6
    assert np.allclose(v2.weights.numpy(), 3 * np.minimum(c.numpy() / 32.0, 1.0), atol=1e-6)


# ---------------------------------------------------------------------------------------------
# global-optimiser edge: gradients and the ray loss of the reference
# (tests/golden/make_golden_grad.py)
# ---------------------------------------------------------------------------------------------
@pytest.fixture()
def insertion_volume():
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    pos = {tuple(k): i for i, k in enumerate(z["keys_sorted"].tolist())}
    perm = np.array([pos[tuple(k)] for k in z["keys_insertion"].tolist()])
    vol = _vol(z)
    vol.insert(torch.from_numpy(z["keys_insertion"]), torch.from_numpy(z["features_sorted"][perm]),
               torch.from_numpy(z["weights_sorted"][perm]), torch.from_numpy(z["num_hits_sorted"][perm]))
    vol.to_tensor()
    vol.features.requires_grad_(True)
    return vol


@pytest.mark.parametrize("name,key", [("random", "random_coords"), ("lattice", "lattice_coords")])
def test_decode_pts_gradient_matches_reference(sd, insertion_volume, name, key):
    vol = insertion_volume
    dec = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    gr = np.load(os.path.join(GOLDEN, "decode_grad_64.npz"))
    out = vol.decode_pts(torch.from_numpy(dec[key]), sd, torch.from_numpy(dec["sdf_delta"]), is_coords=True)
    assert np.abs(out.detach().numpy() - gr[name + "_sdf"]).max() <= FLOAT_TOL
    (out * torch.from_numpy(gr[name + "_grad_out"])).sum().backward()
    ref = gr[name + "_grad_features"]
    assert np.abs(vol.features.grad.numpy() - ref).max() <= 1e-5 * np.abs(ref).max()


def test_calculate_loss_matches_reference(sd, insertion_volume):
    vol = insertion_volume
    dec = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    op = np.load(os.path.join(GOLDEN, "optimize_64.npz"))
    rays = {k[5:]: torch.from_numpy(op[k]) for k in op.files if k.startswith("rays_")}
    gen = torch.Generator().manual_seed(int(op["seed"]))
    loss, pts = orc.calculate_loss(vol, rays, sd, int(op["truncated_units"]), float(op["truncated_dist"]),
                                   int(op["ray_max_dist"]), sdf_delta=torch.from_numpy(dec["sdf_delta"]),
                                   rand=lambda *s: torch.rand(*s, generator=gen))
    assert np.abs(pts.numpy() - op["pts"]).max() <= 1e-7
    assert np.array_equal(vol.weights.numpy(), op["weights_after"])          # count_optim
    loss = loss["depth_bce_loss"]
    assert abs(float(loss.detach()) - float(op["depth_bce_loss"])) <= 1e-6
    loss.backward()
    ref = op["grad_features"]
    assert np.abs(vol.features.grad.numpy() - ref).max() <= 1e-5 * np.abs(ref).max()


def test_tsdf_integrate_matches_reference_cpu_path():
    """tests/golden/tsdf_40.npz: three frames through the reference's own CPU fallback of TSDFVolume.integrate
    (make_golden_tsdf.py).  The oracle's cpu_path flavour reproduces it; the default (CUDA-kernel) flavour differs
    only where the projection of a voxel centre falls within rounding distance of a pixel boundary."""
    z = np.load(os.path.join(GOLDEN, "tsdf_40.npz"))
    dims = z["tsdf"].shape
    res = {}
    for cpu in (True, False):
        tsdf = np.full(dims, -5 * 0.025, dtype=np.float32)
        w = np.zeros(dims, dtype=np.float32)
        for d, T in zip(z["depths"], z["poses"]):
            orc.tsdf_integrate(tsdf, w, z["origin"], float(z["voxel_size"]), d, z["intr"], T, cpu_path=cpu)
        res[cpu] = (tsdf, w)
    assert np.array_equal(res[True][1], z["weight"])
    assert np.abs(res[True][0] - z["tsdf"]).max() <= 2e-7
    observed = int((z["weight"] > 0).sum())
    off = np.abs(res[False][0] - z["tsdf"]) > 1e-6
    assert observed > 10000 and off.sum() <= 0.015 * observed          # pixel-boundary ties only
    assert np.abs(res[False][0] - z["tsdf"])[~off].max() <= 1e-6


def test_depth_front_end_points_match_reference_functions():
    """tests/golden/frontend_120.npz: world points through the reference's geometry.depth2xyz + get_homogeneous as
    FusionInferenceAbstractDataset.__getitem__ calls them (make_golden_frontend.py).  The normals stay unpinned
    (kornia absent)."""
    z = np.load(os.path.join(GOLDEN, "frontend_120.npz"))
    p = orc.depth_to_input_pts(z["depth"], z["intr"], z["T_wc"], float(z["max_depth"]))
    assert p.shape == (int(z["n_valid"]), 6)
    assert np.abs(p[:, :3] - z["pts_w"]).max() <= 1e-15
    assert np.array_equal(p[:, :3].astype(np.float32), z["pts_w"].astype(np.float32))   # what run_e2e.py:249 feeds on


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_headline_config_oracle_vs_reference_golden():
    """The oracle at the configuration the metric is quoted on (256^3, voxel 0.01, full 640x480 frame) against what
    the reference itself produced there (tests/golden/headline_256.npz, make_golden_256.py): encode of frame 0
    (every voxel id and count through SHA-256, features of every 16th voxel), and the lattice decode of 2,048 voxels
    from the reference's own fused volume values after 20 frames."""
    from bnv_fusion_amd import synthetic
    z = np.load(os.path.join(GOLDEN, "headline_256.npz"))
    voxel, dims = float(z["voxel_size"]), z["dims"]
    sd = orc.load_weights(WEIGHTS_FP32)
    torch.set_num_threads(8)
    # inputs are regenerated, not stored: they must be the very frames the reference saw
    assert _sha(synthetic.depth_u16(0)) == str(z["depth_sha256"][0])
    pts = synthetic.frame(0)
    assert _sha(pts) == str(z["input_pts_sha256"][0])
    vol = orc.OracleSparseVolume(8, voxel, dims, 8)
    assert vol.n_xyz.tolist() == [256, 256, 256]
    with torch.no_grad():
        f, c, ids, g, n = orc.encode_pointcloud(sd, torch.from_numpy(pts), vol.n_xyz, vol.min_coords, vol.max_coords,
                                                voxel)
    assert np.array_equal(ids.numpy(), np.cumsum(z["flat_ids_delta_0"].astype(np.int64)))
    assert np.array_equal(c.numpy().reshape(-1), z["pcounts_0"].astype(np.int64))
    assert _sha(ids.numpy().astype(np.int64)) + _sha(c.numpy().reshape(-1).astype(np.int64)) == \
        str(z["ids_counts_sha256"][0])
    assert float(n) == float(z["n_avg_pts"][0])
    assert np.abs(f.numpy()[::16] - z["feats16_0"]).max() <= 2e-6
    # decode from the reference's fused values
    keys = torch.from_numpy(z["nbr_keys"].astype(np.int64))
    vol.insert(keys, torch.from_numpy(z["nbr_feats"]), torch.from_numpy(z["nbr_weights"])[:, None],
               torch.zeros(len(keys), 1))
    origins = z["decode_origins"].astype(np.int64)
    with torch.no_grad():
        got = vol.decode_pts(orc.lattice_coords(origins), sd, None, is_coords=True, query_tensor=False)[0, :, :, 0]
    ref = z["decode_sdf"]
    assert np.array_equal(got.numpy() == np.float32(voxel), ref == np.float32(voxel))      # mask decisions
    assert np.abs(got.numpy() - ref).max() <= 2e-6
    assert (ref != np.float32(voxel)).mean() > 0.3


def test_sweep_window_oracle_vs_reference_golden():
    """The oracle on the moving-camera window the reference itself ran (tests/golden/sweep_256.npz,
    make_golden_sequence.py): the frames in which the camera walks out of the volume -- fewer and fewer points inside,
    points inside but no voxel reaching min_pts, not a single point inside (the reference's `None`) -- ids / counts
    through SHA-256, n_avg_pts; then the lattice decode of 1,024 voxels from the reference's own fused values."""
    from bnv_fusion_amd import sequence, synthetic
    z = np.load(os.path.join(GOLDEN, "sweep_256.npz"))
    voxel, dims = float(z["voxel_size"]), z["dims"]
    H, W = [int(v) for v in z["hw"]]
    scale = sequence.DIMS["golden"][2]
    sd = orc.load_weights(WEIGHTS_FP32)
    torch.set_num_threads(8)
    vol = orc.OracleSparseVolume(8, voxel, dims, 8)
    assert vol.n_xyz.tolist() == [256, 256, 256]
    K = sequence.intrinsics(H, W)
    seen = set()
    for k in range(3, 13):          # t = 445 .. 490: 738 emitted voxels down to none, then no point inside
        t = int(z["frames"][k])
        d16 = sequence.depth_u16(t, H, W, scale, device="cpu").numpy()
        assert _sha(d16) == str(z["depth_sha256"][k])             # the very frames the reference saw
        pts = synthetic.depth_to_input_pts(d16.astype(np.float64) / 1000.0, K, sequence.sweep_pose(t, scale),
                                           max_depth=3.0).astype(np.float32)[None]
        assert _sha(pts) == str(z["input_pts_sha256"][k])
        with torch.no_grad():
            f, c, ids, g, n = orc.encode_pointcloud(sd, torch.from_numpy(pts), vol.n_xyz, vol.min_coords,
                                                    vol.max_coords, voxel)
        if int(z["n_out"][k]) == 0 and float(z["n_avg_pts"][k]) < 0:
            assert f is None                                      # local_point_fusion.py:101-102
            seen.add("none")
            continue
        ids_h, c_h = ids.numpy().astype(np.int64), c.numpy().reshape(-1).astype(np.int64)
        assert len(ids_h) == int(z["n_out"][k])
        assert _sha(ids_h) + _sha(c_h) == str(z["ids_counts_sha256"][k]), k
        assert float(n) == float(z["n_avg_pts"][k])
        seen.add("empty_output" if len(ids_h) == 0 else "voxels")
    assert seen == {"none", "empty_output", "voxels"}
    keys = torch.from_numpy(z["nbr_keys"].astype(np.int64))
    vol.insert(keys, torch.from_numpy(z["nbr_feats"]), torch.from_numpy(z["nbr_weights"])[:, None],
               torch.zeros(len(keys), 1))
    origins = z["decode_origins"].astype(np.int64)
    with torch.no_grad():
        got = vol.decode_pts(orc.lattice_coords(origins), sd, None, is_coords=True, query_tensor=False)[0, :, :, 0]
    ref = z["decode_sdf"]
    assert np.array_equal(got.numpy() == np.float32(voxel), ref == np.float32(voxel))      # mask decisions
    assert np.abs(got.numpy() - ref).max() <= 2e-6
    assert (ref != np.float32(voxel)).mean() > 0.2


def test_noncubic_min_pts5_oracle_vs_reference_golden():
    """The oracle on the reference's own run of a NON-CUBIC volume (n_xyz 105 x 67 x 129: three different strides in
    every flatten / unflatten), voxel 0.02, ``min_pts_in_grid`` 5, a scene cut by the bounds on two axes
    (tests/golden/noncubic.npz, make_golden_noncubic.py): 12 frames of encode -> _integrate (ids / counts through
    SHA-256 and in clear, n_avg_pts, features), the fused volume (keys in insertion order, weights bit-exact), the
    lattice decode of 384 voxels -- a fifth of the rows have a weight in [5, 8): usable here, masked at the default."""
    from bnv_fusion_amd import synthetic
    z = np.load(os.path.join(GOLDEN, "noncubic.npz"))
    voxel, dims, min_pts = float(z["voxel_size"]), z["dims"], int(z["min_pts"])
    H, W = [int(v) for v in z["hw"]]
    sd = orc.load_weights(WEIGHTS_FP32)
    torch.set_num_threads(8)
    vol = orc.OracleSparseVolume(8, voxel, dims, min_pts)
    assert vol.n_xyz.tolist() == z["n_xyz"].tolist() and len(set(vol.n_xyz.tolist())) == 3
    K = synthetic.intrinsics(H, W)
    for k, t in enumerate(int(t) for t in z["frames"]):
        pts = synthetic.depth_to_input_pts(synthetic.depth_u16(t, H, W).astype(np.float64) / 1000.0, K,
                                           synthetic.pose(t), max_depth=3.0).astype(np.float32)[None]
        assert _sha(pts) == str(z["input_pts_sha256"][k])         # the very points the reference saw
        with torch.no_grad():
            f, c, ids, g, n = orc.encode_pointcloud(sd, torch.from_numpy(pts), vol.n_xyz, vol.min_coords,
                                                    vol.max_coords, voxel, min_pts_in_grid=min_pts)
            orc.integrate(vol, g, f, c)
        ids_h, c_h = ids.numpy().astype(np.int64), c.numpy().reshape(-1).astype(np.int64)
        assert np.array_equal(ids_h, np.cumsum(z[f"flat_ids_delta_{k}"].astype(np.int64))), k
        assert np.array_equal(c_h, z[f"pcounts_{k}"].astype(np.int64)) and int(c_h.min()) == min_pts
        assert _sha(ids_h) + _sha(c_h) == str(z["ids_counts_sha256"][k]), k
        assert float(n) == float(z["n_avg_pts"][k])
        if f"feats8_{k}" in z.files:
            assert np.abs(f.numpy()[::8] - z[f"feats8_{k}"]).max() <= 2e-6, k
    vol.to_tensor()
    assert np.array_equal(vol.active_coordinates.numpy(), z["volume_keys"].astype(np.int64))       # insertion order
    assert np.array_equal(vol.weights.numpy().reshape(-1), z["volume_weights"])                    # bit-exact
    assert np.abs(vol.features.numpy()[::8] - z["volume_feats8"]).max() <= 2e-6
    w = z["volume_weights"]
    assert ((w >= 5) & (w < 8)).mean() > 0.1
    origins = z["decode_origins"].astype(np.int64)
    with torch.no_grad():
        got = vol.decode_pts(orc.lattice_coords(origins), sd, None, is_coords=True, query_tensor=False)[0, :, :, 0]
    ref = z["decode_sdf"]
    assert np.array_equal(got.numpy() == np.float32(voxel), ref == np.float32(voxel))      # mask decisions
    assert np.abs(got.numpy() - ref).max() <= 2e-6
    assert (ref != np.float32(voxel)).mean() > 0.3


def test_tcnn_restatement_fp16_accumulate_sensitivity():
    """The tiny-cuda-nn arithmetic is UNPINNED (CUDA-only fp16 kernels, not in the image).  One detail that cannot be
    verified here: whether FullyFusedMLP keeps its accumulator fragments in half precision [from memory: it does].  The
    HIP kernels and the oracle's restatement accumulate in fp32.  This bounds what that detail can change: the same
    frames through the restatement with the accumulator rounded to fp16 after every 16-deep step of the contraction --
    encoder features move by <= 3e-3 (the tolerance the tcnn GPU tests use against the restatement; an fp16 ulp of a
    feature of magnitude 4), the decoded SDF by <= 1e-4 (the north star's bar), with the same mask decisions."""
    from conftest import WEIGHTS_TCNN
    tsd = orc.load_weights(WEIGHTS_TCNN)
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    dims, voxel = z["dims"], float(z["voxel_size"])
    res = {}
    for acc16 in (False, True):
        enc = orc.tcnn_point_encoder(tsd["pointnet_backbone.model.params"], acc16=acc16)
        geo = orc.tcnn_geo_forward(tsd["nerf.model.params"], acc16=acc16)
        vol = orc.OracleSparseVolume(8, voxel, dims, 8)
        feats = []
        with torch.no_grad():
            for fr in z["frames"]:
                f, c, ids, g, n = orc.encode_pointcloud(None, torch.from_numpy(fr), vol.n_xyz, vol.min_coords,
                                                        vol.max_coords, voxel, encoder=enc)
                feats.append(f)
                orc.integrate(vol, g, f, c)
            vol.to_tensor()
            origins = vol.active_coordinates[(vol.weights[:, 0] >= 8).nonzero()[:, 0][:400]]
            sdf = vol.decode_pts(orc.lattice_coords(origins.numpy()), None, None, is_coords=True, geo=geo)[0, :, :, 0]
        res[acc16] = (feats, sdf, origins)
    (fa, sa, oa), (fb, sb, ob) = res[False], res[True]
    assert torch.equal(oa, ob)
    d_feat = max(float((a - b).abs().max()) for a, b in zip(fa, fb))
    assert 0 < d_feat <= 3e-3, d_feat                       # (measured 1.3e-3 on features of magnitude <= 3.6)
    assert torch.equal(sa == voxel, sb == voxel) and float((sa != voxel).float().mean()) > 0.3
    d_sdf = float((sa - sb).abs().max())
    assert 0 < d_sdf <= 1e-4, d_sdf                         # (measured 2.7e-5 at voxel 0.02)
