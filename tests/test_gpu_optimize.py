"""The global-optimiser edge on the GPU: backward of SparseVolume.decode_pts into
``volume.features`` (bnv_decode_pts_backward), the ray loss of render_utils.py and NeuralMap.optimize --
against gradients / losses captured from the reference (tests/golden/make_golden_grad.py) and the oracle.

Bars: forward SDF 1e-4 absolute (north_star); gradients within 1e-4 of the largest gradient entry
(the backward recomputes the MLP in split-f16 arithmetic, ~22 significant bits; atomics add in any order).
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, WEIGHTS_FP32

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GRAD_REL_TOL = 1e-4


@pytest.fixture(scope="module", params=["split_f16", "fp32_exact"])
def bnv(request):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (no CPU fallback exists)")
    import bnv_fusion_amd
    bnv_fusion_amd.set_mlp_mode(1 if request.param == "split_f16" else 0)
    yield bnv_fusion_amd
    bnv_fusion_amd.set_mlp_mode(1)


@pytest.fixture(scope="module")
def model(bnv):
    return bnv.load_pretrained(device=DEV, voxel_size=0.02)


def _insertion_order_volume(bnv):
    """The fused 64^3 volume of sequence_64.npz with rows in the reference's insertion order (the row
    order of the gradient goldens)."""
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    pos = {tuple(k): i for i, k in enumerate(z["keys_sorted"].tolist())}
    perm = np.array([pos[tuple(k)] for k in z["keys_insertion"].tolist()])
    vol = bnv.SparseVolume(8, float(z["voxel_size"]), z["dims"], 8, device=DEV)
    vol.insert(torch.from_numpy(z["keys_insertion"]).to(DEV), torch.from_numpy(z["features_sorted"][perm]).to(DEV),
               torch.from_numpy(z["weights_sorted"][perm]).to(DEV),
               torch.from_numpy(z["num_hits_sorted"][perm]).to(DEV))
    vol.to_tensor()
    assert np.array_equal(vol.active_coordinates.cpu().numpy(), z["keys_insertion"])
    return vol


@pytest.mark.parametrize("name,key", [("random", "random_coords"), ("lattice", "lattice_coords")])
def test_decode_pts_backward_vs_reference_golden(bnv, model, name, key):
    vol = _insertion_order_volume(bnv)
    dec = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    gr = np.load(os.path.join(GOLDEN, "decode_grad_64.npz"))
    vol.features = torch.nn.Parameter(vol.features)                       # run_e2e.py:114
    c = torch.from_numpy(dec[key]).to(DEV)
    delta = torch.from_numpy(dec["sdf_delta"]).to(DEV)
    sdf = vol.decode_pts(c, model.nerf, delta, is_coords=True, query_tensor=True)
    assert sdf.requires_grad and np.abs(sdf.detach().cpu().numpy() - gr[name + "_sdf"]).max() <= 1e-4
    (sdf * torch.from_numpy(gr[name + "_grad_out"]).to(DEV)).sum().backward()
    g, ref = vol.features.grad.cpu().numpy(), gr[name + "_grad_features"]
    assert g.shape == ref.shape
    assert np.abs(g - ref).max() <= GRAD_REL_TOL * np.abs(ref).max(), np.abs(g - ref).max() / np.abs(ref).max()
    assert np.array_equal(np.abs(g).sum(-1) > 0, np.abs(ref).sum(-1) > 0)   # the same rows receive gradient
    # accumulation across two backward calls, like the ray splits of run_e2e.py:124-160
    sdf2 = vol.decode_pts(c, model.nerf, delta, is_coords=True, query_tensor=True)
    (sdf2 * torch.from_numpy(gr[name + "_grad_out"]).to(DEV)).sum().backward()
    assert np.abs(vol.features.grad.cpu().numpy() - 2 * ref).max() <= 2 * GRAD_REL_TOL * np.abs(ref).max()


def test_decode_pts_without_grad_is_plain_forward(bnv, model):
    vol = _insertion_order_volume(bnv)
    dec = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    c = torch.from_numpy(dec["random_coords"]).to(DEV)
    a = vol.decode_pts(c, model.nerf, None, is_coords=True)
    vol.features = torch.nn.Parameter(vol.features)
    with torch.no_grad():
        b = vol.decode_pts(c, model.nerf, None, is_coords=True)
    d = vol.decode_pts(c, model.nerf, None, is_coords=True)
    assert not a.requires_grad and not b.requires_grad and d.requires_grad
    assert torch.equal(a, b) and torch.equal(a, d.detach())


def test_calculate_loss_vs_reference_golden(bnv, model):
    """One calculate_loss of the reference (render_utils.py:551-590) replayed with its CPU random stream."""
    from bnv_fusion_amd import optimize
    vol = _insertion_order_volume(bnv)
    dec = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    op = np.load(os.path.join(GOLDEN, "optimize_64.npz"))
    rays = {k[5:]: torch.from_numpy(op[k]).to(DEV) for k in op.files if k.startswith("rays_")}
    delta = torch.from_numpy(dec["sdf_delta"]).to(DEV)
    vol.features = torch.nn.Parameter(vol.features)
    gen = torch.Generator().manual_seed(int(op["seed"]))
    args = (int(op["truncated_units"]), float(op["truncated_dist"]), int(op["ray_max_dist"]))
    out = optimize.render_with_rays(vol, rays, model.nerf, delta, *args, generator=gen)
    assert np.abs(out["pts_on_rays"].cpu().numpy() - op["pts"]).max() <= 2e-6
    w_after = vol.weights.detach().cpu().numpy()
    assert (w_after != op["weights_after"]).mean() <= 2e-3                # count_optim: same rows (an ulp of a
    vol.weights.copy_(torch.from_numpy(op["weights_before"]).to(DEV))     # sample can flip a boundary corner)
    gen = torch.Generator().manual_seed(int(op["seed"]))
    loss = optimize.calculate_loss(vol, rays, model.nerf, *args, sdf_delta=delta, generator=gen)["depth_bce_loss"]
    assert abs(float(loss.detach()) - float(op["depth_bce_loss"])) <= 1e-4 * float(op["depth_bce_loss"])
    loss.backward()
    g, ref = vol.features.grad.cpu().numpy(), op["grad_features"]
    assert np.abs(g - ref).max() <= 1e-3 * np.abs(ref).max(), np.abs(g - ref).max() / np.abs(ref).max()


def test_backward_vs_oracle_autograd_random_queries(bnv, model):
    """Denser check against torch autograd through the oracle: 4,000 random queries, world coordinates,
    random upstream gradient spanning 6 orders of magnitude (the unit-seed backward must not underflow)."""
    from oracle import bnv_oracle as orc
    sd = orc.load_weights(WEIGHTS_FP32)
    vol = _insertion_order_volume(bnv)
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    ovol = orc.OracleSparseVolume(8, 0.02, z["dims"], 8)
    ovol.insert(vol.active_coordinates.cpu(), vol.features.cpu(), vol.weights.cpu(), vol.num_hits.cpu())
    ovol.to_tensor()
    g = torch.Generator().manual_seed(5)
    rows = torch.randint(len(z["keys_insertion"]), (4000,), generator=g)
    q = (torch.from_numpy(z["keys_insertion"])[rows].float() + (torch.rand(4000, 3, generator=g) - 0.5) * 1.2)
    q = (q * 0.02 + ovol.min_coords).reshape(1, 500, 8, 3)
    go = torch.randn(1, 500, 8, 1, generator=g) * 10.0 ** torch.randint(-7, 0, (1, 500, 8, 1), generator=g).float()
    ovol.features.requires_grad_(True)
    ref_out = ovol.decode_pts(q, sd, None, is_coords=False, query_tensor=True)
    (ref_out * go).sum().backward()
    vol.features = torch.nn.Parameter(vol.features)
    out = vol.decode_pts(q.to(DEV), model.nerf, None, is_coords=False, query_tensor=True)
    (out * go.to(DEV)).sum().backward()
    assert (out.detach().cpu() - ref_out.detach()).abs().max() <= 1e-4
    ref, got = ovol.features.grad, vol.features.grad.cpu()
    assert float((ref.abs().sum(-1) > 0).float().mean()) > 0.3
    assert (got - ref).abs().max() <= GRAD_REL_TOL * ref.abs().max()
    # per-row relative check on rows with a non-negligible gradient
    big = ref.abs().amax(-1) > 1e-3 * ref.abs().max()
    rel = (got[big] - ref[big]).abs().amax(-1) / ref[big].abs().amax(-1)
    assert rel.max() <= 1e-3, rel.max()


def test_neural_map_optimize_reduces_ray_loss(bnv):
    """run_e2e.py:111-162 end to end on synthetic depth frames: fuse, optimise, check that the ray loss on
    held-out samples of the same frames drops and that the optimised features are back in the hash volume."""
    from bnv_fusion_amd import optimize, synthetic
    dims, voxel = synthetic.GRID_DIMS[128]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=200000, device=DEV, tsdf=True)
    H, W = 240, 320
    for t in range(0, 24, 2):
        frame = {"depth": torch.from_numpy(synthetic.depth_u16(t, H, W)).to(DEV),
                 "intr_mat": synthetic.intrinsics(H, W), "T_wc": synthetic.pose(t)}
        nm.integrate(frame)
        nm.frames.append(frame)
    vol = nm.volume
    delta = nm.prepare_tsdf_volume()

    def held_out_loss():
        gen = torch.Generator().manual_seed(99)
        tot = 0.0
        with torch.no_grad():
            for f in nm.frames[::4]:
                rays = optimize.sample_key_frame(f["depth"].to(torch.float32) / 1000.0, f["intr_mat"], f["T_wc"],
                                                 1500, 3, gen)
                tot += float(optimize.calculate_loss(vol, rays, model.nerf, nm.truncated_units, nm.truncated_dist,
                                                     3, sdf_delta=delta, generator=gen)["depth_bce_loss"])
        return tot

    vol.to_tensor()
    w0 = vol.weights.clone()
    before = held_out_loss()
    vol.weights.copy_(w0)
    f0 = vol.features.clone()
    hist = nm.optimize(n_iters=60, last_frame=-1, sampling_size=2000, train_ray_splits=1000, ray_max_dist=3,
                       generator=torch.Generator().manual_seed(1))
    assert len(hist) == 60 and all(torch.isfinite(h) for h in hist)
    after = held_out_loss()
    assert after < 0.99 * before, (before, after)   # lr 1e-3 (run_e2e.py:118): a few % in 60 steps
    assert not vol.features.requires_grad and float((vol.features - f0).abs().max()) > 1e-4
    fq, _, _ = vol.query(vol.active_coordinates[:1000])
    assert torch.equal(fq, vol.features[:1000])                       # run_e2e.py:158-162 write-back


def test_tcnn_decoder_backward_vs_oracle_autograd():
    """decode_pts backward with the reference's default (tiny-cuda-nn, fp16) decoder: MLP mode 2.  PARITY
    UNPINNED like the tcnn forward; checked against torch autograd through the oracle's fp16 restatement of the
    FullyFusedMLP (rounding treated as identity in the backward pass), at fp16-level tolerance."""
    import bnv_fusion_amd as bnv
    from conftest import WEIGHTS_TCNN
    from oracle import bnv_oracle as orc
    model = bnv.load_pretrained(device=DEV, voxel_size=0.02, tiny_cuda=True)
    params = torch.as_tensor(orc.load_weights(WEIGHTS_TCNN)["nerf.model.params"]).float()
    geo = orc.tcnn_geo_forward(params, ste=True)
    vol = _insertion_order_volume(bnv)
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    ovol = orc.OracleSparseVolume(8, 0.02, z["dims"], 8)
    ovol.insert(vol.active_coordinates.cpu(), vol.features.cpu(), vol.weights.cpu(), vol.num_hits.cpu())
    ovol.to_tensor()
    g = torch.Generator().manual_seed(8)
    rows = torch.randint(len(z["keys_insertion"]), (2400,), generator=g)
    q = (torch.from_numpy(z["keys_insertion"])[rows].float() + (torch.rand(2400, 3, generator=g) - 0.5) * 1.2)
    q = q.reshape(1, 300, 8, 3)
    go = torch.randn(1, 300, 8, 1, generator=g)
    ovol.features.requires_grad_(True)
    ref_out = ovol.decode_pts(q, None, None, is_coords=True, query_tensor=True, geo=geo)
    (ref_out * go).sum().backward()
    try:
        vol.features = torch.nn.Parameter(vol.features)
        out = vol.decode_pts(q.to(DEV), model.nerf, None, is_coords=True, query_tensor=True)
        (out * go.to(DEV)).sum().backward()
    finally:
        bnv.set_mlp_mode(1)
    assert (out.detach().cpu() - ref_out.detach()).abs().max() <= 1e-4
    ref, got = ovol.features.grad, vol.features.grad.cpu()
    assert float((ref.abs().sum(-1) > 0).float().mean()) > 0.3
    assert torch.equal(ref.abs().sum(-1) > 0, got.abs().sum(-1) > 0)
    assert (got - ref).abs().max() <= 2e-2 * ref.abs().max(), (got - ref).abs().max() / ref.abs().max()


def test_neural_map_optimize_runs_with_tcnn_checkpoint():
    """The reference's default configuration end to end: tcnn encoder/decoder, fuse, a few optimiser steps."""
    import bnv_fusion_amd as bnv
    from bnv_fusion_amd import synthetic
    dims, voxel = synthetic.GRID_DIMS[128]
    try:
        model = bnv.load_pretrained(device=DEV, voxel_size=voxel, tiny_cuda=True)
        nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=200000, device=DEV, tsdf=True)
        for t in range(0, 12, 2):
            frame = {"depth": torch.from_numpy(synthetic.depth_u16(t, 240, 320)).to(DEV),
                     "intr_mat": synthetic.intrinsics(240, 320), "T_wc": synthetic.pose(t)}
            nm.integrate(frame)
            nm.frames.append(frame)
        nm.volume.to_tensor()
        f0 = nm.volume.features.clone()
        hist = nm.optimize(n_iters=5, sampling_size=1500, train_ray_splits=1000,
                           generator=torch.Generator().manual_seed(2))
        assert len(hist) == 5 and all(torch.isfinite(h) for h in hist)
        assert float((nm.volume.features - f0).abs().max()) > 1e-5
    finally:
        bnv.set_mlp_mode(1)


def test_fused_ray_split_equals_the_torch_formulation(bnv, model):
    """csrc/rays.hip (sampling + loss, fused) + decode forward/backward against optimize.calculate_loss + autograd
    (itself pinned to the reference's golden vectors): same points, same loss, same gradient, same count_optim."""
    from bnv_fusion_amd import optimize
    dec = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    op = np.load(os.path.join(GOLDEN, "optimize_64.npz"))
    rays = {k[5:]: torch.from_numpy(op[k]).to(DEV) for k in op.files if k.startswith("rays_")}
    delta = torch.from_numpy(dec["sdf_delta"]).to(DEV)
    args = (int(op["truncated_units"]), float(op["truncated_dist"]), int(op["ray_max_dist"]))
    vol = _insertion_order_volume(bnv)
    vol.features = torch.nn.Parameter(vol.features)
    grad = torch.zeros_like(vol.features)
    loss, pts = optimize.ray_split_step(vol, rays, model.nerf, *args, sdf_delta=delta,
                                        generator=torch.Generator().manual_seed(int(op["seed"])), grad=grad)
    assert np.abs(pts.cpu().numpy() - op["pts"][0]).max() <= 2e-6
    assert (vol.weights.detach().cpu().numpy() != op["weights_after"]).mean() <= 2e-3
    assert abs(float(loss) - float(op["depth_bce_loss"])) <= 1e-4 * float(op["depth_bce_loss"])
    ref = op["grad_features"]
    assert np.abs(grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max()
    # and against the torch path of this package on a fresh volume, same generator state
    vol2 = _insertion_order_volume(bnv)
    vol2.features = torch.nn.Parameter(vol2.features)
    l2 = optimize.calculate_loss(vol2, rays, model.nerf, *args, sdf_delta=delta,
                                 generator=torch.Generator().manual_seed(int(op["seed"])))["depth_bce_loss"]
    l2.backward()
    assert abs(float(loss) - float(l2.detach())) <= 1e-5 * float(l2.detach())
    assert (grad - vol2.features.grad).abs().max() <= 1e-4 * vol2.features.grad.abs().max()
    assert torch.equal(vol.weights, vol2.weights)


def _split_rays(rays, lo, hi):
    whole = ("T_wc", "intr_mat", "T_wc_host", "intr_host")
    return {k: (v if k in whole else v[:, lo:hi]) for k, v in rays.items()}


@pytest.mark.parametrize("with_delta", [False, True])
def test_batched_step_equals_split_by_split(bnv, model, with_delta):
    """optimize.ray_batch_step (bnv_optim_step: ALL ray splits of a step in one forward + loss + backward launch, the
    count_optim of the splits deferred as per-row split masks) against the split-by-split sequence of ray_split_step
    (count_optim -> decode_pts -> loss -> backward per split, the reference's order run_e2e.py:127-153): same sample
    points, the same mask decision for every sample -- with weights set so that a corner only goes live once TWO splits
    have touched it, i.e. the decisions differ from split to split --, same values, same loss, same gradient, and the
    same weights afterwards, bit for bit."""
    from bnv_fusion_amd import optimize
    dec = np.load(os.path.join(GOLDEN, "decode_64.npz"))
    op = np.load(os.path.join(GOLDEN, "optimize_64.npz"))
    rays = {k[5:]: torch.from_numpy(op[k]).to(DEV) for k in op.files if k.startswith("rays_")}
    delta = torch.from_numpy(dec["sdf_delta"]).to(DEV) if with_delta else None
    args = (int(op["truncated_units"]), float(op["truncated_dist"]), int(op["ray_max_dist"]))
    n, per = int(rays["uv"].shape[1]), 40            # 4 splits of 40 rays x 35 samples (1,400: not a multiple of a chunk)
    vols = []
    for _ in range(2):
        v = _insertion_order_volume(bnv)
        v.weights.copy_(torch.clamp(v.weights, max=6.5))      # live needs 6.5 + 1 + 1: two splits must touch a corner
        v.features = torch.nn.Parameter(v.features)
        vols.append(v)
    va, vb = vols
    # A: split by split
    gen = torch.Generator().manual_seed(7)
    grad_a = torch.zeros_like(va.features)
    loss_a, pts_a, pred_a = [], [], []
    for lo in range(0, n, per):
        l, p, q = optimize.ray_split_step(va, _split_rays(rays, lo, lo + per), model.nerf, *args, sdf_delta=delta,
                                          generator=gen, grad=grad_a, return_pred=True)
        loss_a.append(l)
        pts_a.append(p)
        pred_a.append(q)
    pts_a, pred_a = torch.cat(pts_a), torch.cat(pred_a)
    # B: all splits at once
    grad_b = torch.zeros_like(vb.features)
    loss_b, pts_b, pred_b = optimize.ray_batch_step(vb, rays, model.nerf, *args, sdf_delta=delta,
                                                    generator=torch.Generator().manual_seed(7), grad=grad_b,
                                                    train_ray_splits=per, return_pred=True)
    assert torch.equal(pts_a, pts_b)
    assert torch.equal(va.weights, vb.weights)                       # count_optim: same +1s, same float sums
    assert not vb._split_mask().any()                                # the masks are clear for the next step
    voxel = float(va.voxel_size)
    masked_a = pred_a == voxel if delta is None else None
    if masked_a is not None:
        masked_b = pred_b == voxel
        assert torch.equal(masked_a, masked_b)                       # every mask decision
        live = (~masked_a).view(n // per, -1).float().mean(1)
        assert float(live[0]) == 0.0 and float(live[-1]) > 0.02 and float(live[1]) > 0.0, live   # they DO differ by split
    assert float((pred_a - pred_b).abs().max()) <= 2e-7
    la = float(sum(float(x) for x in loss_a))
    assert abs(float(loss_b) - la) <= 2e-6 * abs(la)
    assert float(grad_a.abs().max()) > 0
    assert float((grad_a - grad_b).abs().max()) <= 2e-5 * float(grad_a.abs().max())


def test_batched_optimize_follows_the_split_by_split_optimiser(bnv):
    """NeuralMap.optimize through optimize_volume(batched=True) -- one fused launch per step -- against batched=False
    (round 5's split-by-split step) from the same generator: the same losses step for step (to float-atomics order) and
    the same final weights (count_optim) bit for bit."""
    from bnv_fusion_amd import optimize, synthetic
    dims, voxel = synthetic.GRID_DIMS[128]
    outs = []
    for batched in (True, False):
        model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
        nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=200000, device=DEV, tsdf=False)
        H, W = 240, 320
        for t in range(0, 20, 2):
            frame = {"depth": torch.from_numpy(synthetic.depth_u16(t, H, W)).to(DEV),
                     "intr_mat": synthetic.intrinsics(H, W), "T_wc": synthetic.pose(t)}
            nm.integrate(frame)
            nm.frames.append(frame)
        gen = torch.Generator().manual_seed(3)

        def batches():
            for i in range(6):
                f = nm.frames[i % len(nm.frames)]
                yield optimize.sample_key_frame(f["depth"].float() / 1000.0, f["intr_mat"], f["T_wc"], 2500, 3, gen)

        hist = optimize.optimize_volume(nm.volume, model.nerf, batches(), nm.truncated_units, nm.truncated_dist, 3,
                                        train_ray_splits=500, generator=gen, batched=batched)
        outs.append((torch.stack([h.reshape(()) for h in hist]).cpu(), nm.volume.weights.clone(),
                     nm.volume.features.clone()))
    (ha, wa, fa), (hb, wb, fb) = outs
    assert torch.equal(wa, wb)
    assert float((ha[0] - hb[0]).abs() / hb[0].abs()) <= 1e-5, (ha, hb)      # the first step: identical inputs
    # later steps: Adam's first updates are +-lr whatever a gradient's size, so an entry whose gradient is rounding
    # noise around zero (float atomics sum in no fixed order, in either path) can step the other way
    assert float(((ha - hb).abs() / hb.abs()).max()) <= 2e-3, (ha, hb)
    assert float((fa - fb).abs().max()) <= 1.3e-2 and float((fa - fb).abs().mean()) <= 1e-4
