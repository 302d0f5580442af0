"""The f16-split MLP arithmetic (modes 1 and 3) at the edges of its range, against the fp32 oracle: tiny operands
(subnormal ``lo`` halves), large operands (``hi`` halves near the f16 maximum), and beyond the certified range, where
the kernels must raise a sticky error instead of silently producing inf / NaN (fp32, the reference's arithmetic,
handles those inputs; exact mode 0 must too).  Needs a real MI355X: run with  -m gpu."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, WEIGHTS_FP32

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def env():
    import bnv_fusion_amd as bnv
    from bnv_fusion_amd import weights
    from oracle import bnv_oracle as orc
    sd = orc.load_weights(WEIGHTS_FP32)
    model = bnv.load_pretrained(device=DEV, voxel_size=0.02)
    yield bnv, orc, sd, model, weights
    bnv.set_mlp_mode(1)


def _volumes(bnv, orc, scale):
    z = np.load(os.path.join(GOLDEN, "sequence_64.npz"))
    voxel, dims = float(z["voxel_size"]), z["dims"]
    keys = torch.from_numpy(z["keys_sorted"])
    f = torch.from_numpy(z["features_sorted"]) * scale
    w = torch.from_numpy(z["weights_sorted"])
    vol = bnv.SparseVolume(8, voxel, dims, 8, device=DEV)
    vol.insert(keys.to(DEV), f.to(DEV), w.to(DEV), torch.zeros(len(keys), 1, device=DEV))
    ovol = orc.OracleSparseVolume(8, voxel, dims, 8)
    ovol.insert(keys, f, w, torch.zeros(len(keys), 1))
    live = keys[w[:, 0] >= 8]
    return vol, ovol, live[:: max(1, len(live) // 300)][:300], voxel


@pytest.mark.parametrize("scale", [1e-6, 1e-3, 1.0, 30.0, 100.0])
@pytest.mark.parametrize("mode", [1, 3])
def test_decode_inside_the_certified_range(env, scale, mode):
    """Feature rows scaled from 1e-6 (f16-subnormal hi / vanishing lo) to 100 (|feature| up to ~300 of the ~380
    certified; hidden activations in the thousands): the split arithmetic stays fp32-class (mode 1) or
    f16-operand-class (mode 3) against the oracle, mask decisions identical, no error raised."""
    bnv, orc, sd, model, weights = env
    assert 300.0 < weights.sdf_feature_bound(sd) < 1e4
    vol, ovol, pick, voxel = _volumes(bnv, orc, scale)
    assert float(vol._features.abs().max()) < weights.sdf_feature_bound(sd)
    with torch.no_grad():
        ref = ovol.decode_pts(orc.lattice_coords(pick.numpy()), sd, None, is_coords=True, query_tensor=False)[0, :, :, 0]
    bnv.set_mlp_mode(mode)
    got = vol.decode_lattice(pick.to(DEV), model.nerf, query_tensor=False).cpu()
    pts = vol.decode_pts(orc.lattice_coords(pick.numpy()[:64]).to(DEV), model.nerf, None, is_coords=True,
                         query_tensor=False).cpu()[0, :, :, 0]
    vol.num_rows()                                     # raises if a kernel flagged the range
    tol = (1e-4 if mode == 1 else 3e-3) * max(1.0, float(ref.abs().max()))
    assert torch.equal(got == voxel, ref == voxel) and float((ref != voxel).float().mean()) > 0.3
    assert float((got - ref).abs().max()) <= tol, (scale, float((got - ref).abs().max()), tol)
    assert float((pts - ref[:64]).abs().max()) <= tol
    if mode == 1 and scale <= 1.0:
        assert float((got - ref).abs().max()) <= 1e-4              # the north-star bar, to the letter


@pytest.mark.parametrize("scale", [1e4, float("nan")])
def test_decode_beyond_the_certified_range_is_flagged(env, scale):
    """|feature| beyond the certificate (hidden activations would pass 65,504) or NaN: the split modes raise the
    volume's sticky error word; exact fp32 decodes the same volume like the oracle does."""
    bnv, orc, sd, model, weights = env
    s = 1e4 if scale != scale else scale
    vol, ovol, pick, voxel = _volumes(bnv, orc, s)
    if scale != scale:                                 # one NaN feature in a live row
        f, w, _ = vol.query(pick[:1].to(DEV))
        f[0, 3] = float("nan")
        vol.insert(pick[:1].to(DEV), f, w, torch.zeros(1, 1, device=DEV))
    for mode in (1, 3):
        bnv.set_mlp_mode(mode)
        vol.decode_lattice(pick.to(DEV), model.nerf, query_tensor=False)
        with pytest.raises(bnv.BnvError, match="certified"):
            vol.num_rows()
        vol._status[1] = 0                             # acknowledge
    if scale == scale:
        bnv.set_mlp_mode(0)
        with torch.no_grad():
            ref = ovol.decode_pts(orc.lattice_coords(pick.numpy()), sd, None, is_coords=True, query_tensor=False)[0, :, :, 0]
        got = vol.decode_lattice(pick.to(DEV), model.nerf, query_tensor=False).cpu()
        vol.num_rows()
        assert torch.isfinite(got).all() and float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_encoder_normal_range(env):
    """Point encoder: |normal| up to 100 (certified bound ~144) stays fp32-class in split mode; |normal| = 1e3 is
    flagged through the frame's error word in split mode and encoded correctly in exact fp32."""
    bnv, orc, sd, model, weights = env
    assert 100.0 < weights.pointnet_normal_bound(sd) < 1e3
    z = np.load(os.path.join(GOLDEN, "encode_64.npz"))
    vol = bnv.SparseVolume(8, float(z["voxel_size"]), z["dims"], 8, device=DEV)
    ovol = orc.OracleSparseVolume(8, float(z["voxel_size"]), z["dims"], 8)
    base = torch.from_numpy(z["input_pts"]).clone()

    def run(mode, nscale):
        pts = base.clone()
        pts[..., 3:] *= nscale
        bnv.set_mlp_mode(mode)
        out = model.encode_pointcloud(pts.to(DEV), vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size,
                                      return_dense=False)
        with torch.no_grad():
            ref = orc.encode_pointcloud(sd, pts, ovol.n_xyz, ovol.min_coords, ovol.max_coords, ovol.voxel_size)
        assert torch.equal(out[2].cpu(), ref[2]) and torch.equal(out[1].cpu(), ref[1])
        return float((out[0].cpu() - ref[0]).abs().max()), float(ref[0].abs().max())

    for nscale in (1e-6, 1.0, 100.0):
        err, mag = run(1, nscale)
        assert err <= 1e-4 * max(1.0, mag), (nscale, err, mag)
    with pytest.raises(bnv.BnvError, match="certified"):
        run(1, 1e3)
    err, mag = run(0, 1e3)
    assert err <= 1e-4 * max(1.0, mag)
