"""Weights of the fp32 point encoder / SDF decoder: loading, BatchNorm folding and the
pre-permuted MFMA operand layouts the kernels read (csrc/encode.hip, csrc/decode.hip).

State-dict key names are the reference's, so either the converted
``weights/pointnet_fp32.npz`` or a ``torch.load(ckpt)['state_dict']`` can be passed in.
"""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_FP32 = os.path.join(_HERE, "weights", "pointnet_fp32.npz")
DEFAULT_TCNN = os.path.join(_HERE, "weights", "pointnet_tcnn.npz")

BN_EPS = 1e-5  # nn.BatchNorm1d default (pointnet_utils.py:240-243)


def load_npz(path=DEFAULT_FP32):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def _np(v):
    return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)


def fold_pointnet(sd):
    """conv1d(k=1) + eval-mode BatchNorm1d -> one affine map per layer (float64 fold).
    pointnet_utils.py:246-266: y = (W x + b - mean) / sqrt(var + eps) * gamma + beta."""
    out = []
    for i in (1, 2, 3, 4):
        p = "pointnet_backbone."
        W = _np(sd[f"{p}conv{i}.weight"]).astype(np.float64)[:, :, 0]
        b = _np(sd[f"{p}conv{i}.bias"]).astype(np.float64)
        s = _np(sd[f"{p}bn{i}.weight"]).astype(np.float64) / np.sqrt(
            _np(sd[f"{p}bn{i}.running_var"]).astype(np.float64) + BN_EPS)
        Wf = W * s[:, None]
        bf = (b - _np(sd[f"{p}bn{i}.running_mean"]).astype(np.float64)) * s + _np(sd[f"{p}bn{i}.bias"]).astype(
            np.float64)
        out.append((Wf.astype(np.float32), bf.astype(np.float32)))
    return out


F16_MAX = 65504.0


def certified_input_bound(layers, fixed, n_free, limit=0.98 * F16_MAX):
    """Range certificate of the f16-split arithmetic (MLP modes 1 and 3): every value a layer hands to the next one
    is rounded to an f16 ``hi`` part, which overflows at 65,520 -- IEEE fp32, the reference's arithmetic, does not.
    Interval bound: with |input_j| <= fixed_j for the first inputs and <= B for the last ``n_free`` ones, the
    pre-activations of layer l are bounded per neuron by a_l + B c_l with a_0 = |W_0| fixed + |b_0|,
    c_0 = |W_0| 1_free, a_l = |W_l| a_{l-1} + |b_l|, c_l = |W_l| c_{l-1} (ReLU only shrinks magnitudes).  Returns
    the largest B for which every layer passed in stays below ``limit``: inputs within that bound CANNOT overflow.
    ``layers``: the (W, b) pairs whose outputs are converted to f16 (all but the last layer of a network)."""
    W0 = np.abs(np.asarray(layers[0][0], dtype=np.float64))
    fixed = np.asarray(fixed, dtype=np.float64)
    k = len(fixed)
    assert W0.shape[1] == k + n_free
    a = W0[:, :k] @ fixed + np.abs(np.asarray(layers[0][1], dtype=np.float64))
    c = W0[:, k:].sum(1)
    best = np.inf
    for i, (W, b) in enumerate(layers):
        if i:
            Wa = np.abs(np.asarray(W, dtype=np.float64))
            a, c = Wa @ a + np.abs(np.asarray(b, dtype=np.float64)), Wa @ c
        if (a >= limit).any():
            return 0.0
        with np.errstate(divide="ignore"):
            best = min(best, float(np.min(np.where(c > 0, (limit - a) / c, np.inf))))
    return best


def pointnet_normal_bound(sd):
    """Largest |normal component| for which the split-mode point encoder provably cannot overflow (the three
    relative coordinates are within [-1, 1] by construction, local_point_fusion.py:60-61)."""
    return certified_input_bound(fold_pointnet(sd)[:3], [1.0, 1.0, 1.0], 3)


def sdf_feature_bound(sd):
    """Largest |feature| for which the split-mode SDF decoder provably cannot overflow.  Position inputs: local
    coordinates are within [-1, 1]; the global-coordinate branch of decode_feature_grid_w_pts feeds coords / (res - 1),
    at most 1 + 0.5 / (res - 1) <= 1.5 for a query whose nearest voxel is inside the grid (any other query is masked
    to voxel_size whatever the MLP returns) -- the certificate uses 1.5; sin / cos within [-1, 1]; the last hidden
    layer's output is consumed in fp32."""
    layers = [(_np(sd[f"nerf.geo_layer{i}.weight"]), _np(sd[f"nerf.geo_layer{i}.bias"])) for i in range(3)]
    return certified_input_bound(layers, [1.5] * 3 + [1.0] * 6, 8)


def pack_pointnet(sd):
    """-> float32 [34952 + split pack + 4] in the PN_* / PX_* layouts of csrc/encode.hip; the trailing 4 floats hold the
    certified bound on |normal| of the split modes (pointnet_normal_bound) and padding.

    MFMA tile: lane l = (n = l & 31, h = l >> 5).  A K-step that consumes D register r of input
    block nb contracts input features nb*32 + f0(r) + 4h with f0(r) = (r & 3) + 8 (r >> 2); for
    r = 4 rq + i that is nb*32 + 8 rq + i + 4h."""
    (W1, b1), (W2, b2), (W3, b3), (W4, b4) = fold_pointnet(sd)
    lane = np.arange(64)
    n, h = lane & 31, lane >> 5
    # W1p[s][mb][l] = W1[mb*32 + n][2s + h]
    w1p = np.zeros((3, 4, 64), np.float32)
    for s in range(3):
        for mb in range(4):
            w1p[s, mb] = W1[mb * 32 + n, 2 * s + h]

    def pack128(W):
        o = np.zeros((4, 4, 4, 64, 4), np.float32)
        for mb in range(4):
            for nb in range(4):
                for rq in range(4):
                    for i in range(4):
                        o[mb, nb, rq, :, i] = W[mb * 32 + n, nb * 32 + 8 * rq + i + 4 * h]
        return o

    w4p = np.zeros((4, 4, 2, 8, 4), np.float32)
    for nb in range(4):
        for rq in range(4):
            for hh in range(2):
                for i in range(4):
                    w4p[nb, rq, hh, :, i] = W4[:, nb * 32 + 8 * rq + i + 4 * hh]
    fp32_part = np.concatenate([w1p.ravel(), pack128(W2).ravel(), pack128(W3).ravel(), w4p.ravel(),
                                b1, b2, b3, b4]).astype(np.float32)
    trailer = np.zeros(4, np.float32)
    trailer[0] = min(pointnet_normal_bound(sd), 3.0e38)
    return np.concatenate([fp32_part, _pack_pointnet_split16(W1, W2, W3, W4), trailer])


def split_f16(x):
    """fp32 -> (hi, lo) float16 with x ~ hi + lo (about 22 significant bits; lo may be subnormal)."""
    x = np.asarray(x, dtype=np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def _slot_feature(jj, h):
    """Feature (within a 16-deep K-step) held in operand slot jj of lane half h."""
    return 8 * (jj >> 2) + 4 * h + (jj & 3)


def _pack_pointnet_split16(W1, W2, W3, W4):
    """Split-operand layout PX_* of csrc/encode.hip (k_pointnet_scatter_x, v_mfma_f32_16x16x32_f16), as float32 words.
    A fragments: lane (m = l & 15, g = l >> 4), slot jj.  Layer 1: [8 rb][hi/lo][64][8] = W1[16 rb + m][8 g + jj]
    (6 inputs, the other 26 K slots zero).  Layers 2, 3: [4 s][8 rb][hi/lo][64][8] = W[16 rb + m][32 s + 16 (jj >> 2) +
    4 g + (jj & 3)] -- K-step s consumes what accumulator row blocks 2 s and 2 s + 1 of the previous layer hold.
    Layer 4: [4 s][hi/lo][64][8], rows m >= 8 zero."""
    lane = np.arange(64)
    m, g = lane & 15, lane >> 4
    jj = np.arange(8)
    W1p = np.zeros((128, 32), np.float32)
    W1p[:, :6] = W1
    w1 = np.zeros((8, 2, 64, 8), np.float16)
    for rb in range(8):
        w1[rb, 0], w1[rb, 1] = split_f16(W1p[(16 * rb + m)[:, None], 8 * g[:, None] + jj[None, :]])

    def kidx(s):
        return 32 * s + 16 * (jj[None, :] >> 2) + 4 * g[:, None] + (jj[None, :] & 3)

    def pack128(W):
        o = np.zeros((4, 8, 2, 64, 8), np.float16)
        for s in range(4):
            for rb in range(8):
                o[s, rb, 0], o[s, rb, 1] = split_f16(W[(16 * rb + m)[:, None], kidx(s)])
        return o

    W4p = np.zeros((16, 128), np.float32)
    W4p[:8] = W4
    w4 = np.zeros((4, 2, 64, 8), np.float16)
    for s in range(4):
        w4[s, 0], w4[s, 1] = split_f16(W4p[m[:, None], kidx(s)])
    halves = np.concatenate([w1.ravel(), pack128(W2).ravel(), pack128(W3).ravel(), w4.ravel()])
    assert halves.size == 77824
    return halves.view(np.float32)


def _pack_split(W, nks, n_blocks=8):
    """[n_blocks w][nks][hi/lo][64 lane][8] halves: slot jj of lane (n, h) = W[32 w + n][16 ks + slot_feature(jj, h)]
    (rows / columns beyond W's shape are zero)."""
    lane = np.arange(64)
    n, h = lane & 31, lane >> 5
    Wp = np.zeros((32 * n_blocks, 16 * nks), np.float32)
    Wp[:W.shape[0], :W.shape[1]] = W
    jj = np.arange(8)
    o = np.zeros((n_blocks, nks, 2, 64, 8), np.float16)
    for w in range(n_blocks):
        for ks in range(nks):
            v = Wp[(32 * w + n)[:, None], 16 * ks + _slot_feature(jj[None, :], h[:, None])]
            o[w, ks, 0], o[w, ks, 1] = split_f16(v)
    return o.ravel()


def _pack_split16(W, n_steps, first_layer=False):
    """Split pack for v_mfma_f32_16x16x32_f16 (SX_* of csrc/decode.hip, k_lattice_table_x):
    [8 w][2 n_steps units][hi/lo][64 lane][8] halves, unit u = 2 s + rb.  Slot jj of lane (m = l & 15, g = l >> 4) =
    W[32 w + 16 rb + m][k] with k = 32 s + 16 (jj >> 2) + 4 g + (jj & 3) -- the order in which a wave's accumulator
    registers leave the previous layer -- or, for the first layer, k = 8 g + jj (inputs beyond W's columns are 0)."""
    lane = np.arange(64)
    m, g = lane & 15, lane >> 4
    jj = np.arange(8)
    Wp = np.zeros((256, 32 * n_steps), np.float32)
    Wp[:W.shape[0], :W.shape[1]] = W
    o = np.zeros((8, 2 * n_steps, 2, 64, 8), np.float16)
    for w in range(8):
        for s in range(n_steps):
            for rb in range(2):
                if first_layer:
                    k = 8 * g[:, None] + jj[None, :]
                else:
                    k = 32 * s + 16 * (jj[None, :] >> 2) + 4 * g[:, None] + (jj[None, :] & 3)
                v = Wp[(32 * w + 16 * rb + m)[:, None], k]
                o[w, 2 * s + rb, 0], o[w, 2 * s + rb, 1] = split_f16(v)
    return o.ravel()


def pack_sdf_mlp_bwd(sd):
    """Transposed layers for bnv_decode_pts_backward -> float32 [204800] (SB_* layout of csrc/decode.hip):
    W3^T, W2^T, W1^T as 256x256 split packs, then W0^T (17 x 256, rows padded to 32)."""
    Ws = [_np(sd[f"nerf.geo_layer{i}.weight"]).astype(np.float32) for i in range(4)]
    halves = np.concatenate([_pack_split(Ws[3].T, 16), _pack_split(Ws[2].T, 16), _pack_split(Ws[1].T, 16),
                             _pack_split(Ws[0].T, 16, n_blocks=1)])
    assert halves.size == 409600
    return halves.view(np.float32).copy()


def pack_sdf_mlp(sd):
    """-> float32 [SD_TOTAL] in the SD_* layout of csrc/decode.hip.
    Wp[w][kb][l][i] = W[32 w + (l & 31)][8 kb + 4 (l >> 5) + i]; layer 0 has K = 17 padded to 24."""
    lane = np.arange(64)
    n, h = lane & 31, lane >> 5

    def pack(W, nkb):
        K = W.shape[1]
        o = np.zeros((8, nkb, 64, 4), np.float32)
        for w in range(8):
            for kb in range(nkb):
                for i in range(4):
                    k = 8 * kb + 4 * h + i
                    ok = k < K
                    o[w, kb, ok, i] = W[32 * w + n[ok], k[ok]]
        return o.ravel()

    Ws = [_np(sd[f"nerf.geo_layer{i}.weight"]).astype(np.float32) for i in range(4)]
    bs = [_np(sd[f"nerf.geo_layer{i}.bias"]).astype(np.float32) for i in range(4)]
    assert Ws[0].shape == (256, 17) and all(W.shape == (256, 256) for W in Ws[1:])
    wa = _np(sd["nerf.fc_alpha.weight"]).astype(np.float32).reshape(256)
    ba = np.zeros(4, np.float32)
    ba[0] = _np(sd["nerf.fc_alpha.bias"]).reshape(-1)[0]
    ba[1] = min(sdf_feature_bound(sd), 3.0e38)     # certified |feature| bound of the split modes (SD_BA + 1)
    fp32_part = np.concatenate([pack(Ws[0], 3), pack(Ws[1], 32), pack(Ws[2], 32), pack(Ws[3], 32),
                                bs[0], bs[1], bs[2], bs[3], wa, ba]).astype(np.float32)

    halves = np.concatenate([_pack_split(Ws[0], 2), _pack_split(Ws[1], 16), _pack_split(Ws[2], 16),
                             _pack_split(Ws[3], 16)])
    assert halves.size == 409600
    # the same four layers in the operand order of the 16x16x32 MFMA (k_lattice_table_x), behind the 32x32x16 pack
    halves_x = np.concatenate([_pack_split16(Ws[0], 1, first_layer=True), _pack_split16(Ws[1], 8),
                               _pack_split16(Ws[2], 8), _pack_split16(Ws[3], 8)])
    assert halves_x.size == 409600
    return np.concatenate([fp32_part, halves.view(np.float32), halves_x.view(np.float32)])


# --------------------------------------------------------------------------------------------------
# tiny-cuda-nn checkpoints (pointnet_tcnn.ckpt): one flat fp32 master vector per network holding the
# row-major [out, in] matrices of FullyFusedMLP in order.
# --------------------------------------------------------------------------------------------------
def _split_tcnn(params, n_in_padded, width=64, n_hidden=3):
    params = _np(params).astype(np.float32).reshape(-1)
    dims = [n_in_padded] + [width] * n_hidden + [16]
    mats, off = [], 0
    for i in range(len(dims) - 1):
        mats.append(params[off: off + dims[i + 1] * dims[i]].reshape(dims[i + 1], dims[i]))
        off += dims[i + 1] * dims[i]
    assert off == params.size, (off, params.size)
    return mats


def _pack_tcnn(mats):
    """-> float32 words holding f16 fragments: first layer [2 mb][nks][64][8]; hidden layers
    [2 mb][4 g][64][8] with g = (input block nb, ksl); output layer [4 g][64][8] (rows >= 16 zero)."""
    lane = np.arange(64)
    n, h = lane & 31, lane >> 5
    jj = np.arange(8)
    sf = _slot_feature(jj[None, :], h[:, None])
    out = []
    W0 = mats[0]
    nks = W0.shape[1] // 16
    o = np.zeros((2, nks, 64, 8), np.float16)
    for mb in range(2):
        for ks in range(nks):
            o[mb, ks] = W0[(mb * 32 + n)[:, None], 16 * ks + sf]
    out.append(o.ravel())
    for W in mats[1:-1]:
        o = np.zeros((2, 4, 64, 8), np.float16)
        for mb in range(2):
            for g in range(4):
                o[mb, g] = W[(mb * 32 + n)[:, None], (g >> 1) * 32 + 16 * (g & 1) + sf]
        out.append(o.ravel())
    Wl = np.zeros((32, 64), np.float32)
    Wl[:16] = mats[-1]
    o = np.zeros((4, 64, 8), np.float16)
    for g in range(4):
        o[g] = Wl[n[:, None], (g >> 1) * 32 + 16 * (g & 1) + sf]
    out.append(o.ravel())
    return np.concatenate(out).view(np.float32)


def pack_pointnet_tcnn(params):
    """tcnnPointNetEncoder (pointnet_utils.py:269-294): 16 | 64 | 64 | 64 | 16 -> PT_* layout of csrc/encode.hip."""
    return _pack_tcnn(_split_tcnn(params, 16))


def pack_sdf_tcnn(params):
    """tcnnNeRFModel (modules.py:136-253): 32 | 64 | 64 | 64 | 16 -> ST_* layout of csrc/decode.hip."""
    return _pack_tcnn(_split_tcnn(params, 32))


def pack_sdf_tcnn_bwd(params):
    """Transposed layers of the tcnn SDF decoder for bnv_decode_pts_backward in MLP mode 2 -> float32 [5184]
    (TB_* layout of csrc/decode.hip): W2^T, W1^T as hidden-layer packs, W0^T as an output-layer pack (32 rows =
    the padded inputs), then row 0 of the output layer as 64 floats."""
    W0, W1, W2, W3 = _split_tcnn(params, 32)
    h16 = lambda W: W.astype(np.float16).astype(np.float32)     # the forward multiplies fp16 weights
    lane = np.arange(64)
    n, h = lane & 31, lane >> 5
    sf = _slot_feature(np.arange(8)[None, :], h[:, None])
    out = []
    for W in (h16(W2).T, h16(W1).T):
        o = np.zeros((2, 4, 64, 8), np.float16)
        for mb in range(2):
            for g in range(4):
                o[mb, g] = W[(mb * 32 + n)[:, None], (g >> 1) * 32 + 16 * (g & 1) + sf]
        out.append(o.ravel())
    Wt = h16(W0).T                                                # [32, 64]
    o = np.zeros((4, 64, 8), np.float16)
    for g in range(4):
        o[g] = Wt[n[:, None], (g >> 1) * 32 + 16 * (g & 1) + sf]
    out.append(o.ravel())
    halves = np.concatenate(out).view(np.float32)
    return np.concatenate([halves, h16(W3)[0].astype(np.float32)])
