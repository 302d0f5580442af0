"""CPU emulation of the v_mfma_f32_32x32x2_f32 lane/register layout, used to check that the
host-side weight packing (bnv_fusion_amd/weights.py) matches the index arithmetic of the kernels
(csrc/encode.hip k_pointnet_scatter, csrc/decode.hip sdf_mlp_tile) without a GPU.

Layout (cdna_hip_programming.md section 3): A lane l holds A[i = l & 31][k = l >> 5]; B lane l holds
B[k = l >> 5][j = l & 31]; D lane l register r holds D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31].
"""
import numpy as np
import torch

from bnv_fusion_amd import weights as W
from oracle import bnv_oracle as orc
from conftest import WEIGHTS_FP32

LANE = np.arange(64)
N_, H_ = LANE & 31, LANE >> 5
ROW = np.array([[(r & 3) + 8 * (r >> 2) + 4 * h for r in range(16)] for h in H_])  # [64 lanes][16 regs]


def mfma(a, b, c):
    """a, b: [64] per-lane operands; c: [64, 16] accumulator -> d [64, 16] (float64 emulation)."""
    A = np.zeros((32, 2))
    B = np.zeros((2, 32))
    A[N_, H_] = a
    B[H_, N_] = b
    D = A @ B
    return c + D[ROW, N_[:, None]]


def test_pointnet_pack_matches_direct_mlp():
    sd = W.load_npz(WEIGHTS_FP32)
    pack = W.pack_pointnet(sd).astype(np.float64)
    PN_W1, PN_W2 = 0, 768
    PN_W3 = PN_W2 + 16384
    PN_W4 = PN_W3 + 16384
    PN_B1 = PN_W4 + 1024
    PN_B2, PN_B3, PN_B4 = PN_B1 + 128, PN_B1 + 256, PN_B1 + 384
    assert PN_B4 + 8 == 34952 and pack.size == 34952 + 77824 // 2 + 4      # + 16x16x32 split pack + certified-range trailer
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(32, 6))  # 32 pairs

    def bias_init(off, mb):
        v = np.zeros((64, 16))
        for q in range(4):
            for i in range(4):
                v[:, 4 * q + i] = pack[off + mb * 32 + 8 * q + 4 * H_ + i]
        return v

    # layer 1
    ha = [bias_init(PN_B1, mb) for mb in range(4)]
    for s in range(3):
        b = x[N_, 2 * s + H_]
        for mb in range(4):
            ha[mb] = mfma(pack[PN_W1 + (s * 4 + mb) * 64 + LANE], b, ha[mb])
    ha = [np.maximum(v, 0) for v in ha]

    def layer128(woff, boff, inp):
        out = [bias_init(boff, mb) for mb in range(4)]
        for nb in range(4):
            for rq in range(4):
                for i in range(4):
                    for mb in range(4):
                        a = pack[woff + (((mb * 4 + nb) * 4 + rq) * 64 + LANE) * 4 + i]
                        out[mb] = mfma(a, inp[nb][:, 4 * rq + i], out[mb])
        return out

    hb = [np.maximum(v, 0) for v in layer128(PN_W2, PN_B2, ha)]
    ha = [np.maximum(v, 0) for v in layer128(PN_W3, PN_B3, hb)]
    o = np.zeros((64, 16))
    for r in range(4):
        o[:, r] = pack[PN_B4 + 4 * H_ + r]
    for nb in range(4):
        for rq in range(4):
            for i in range(4):
                a = np.where(N_ < 8, pack[PN_W4 + ((((nb * 4 + rq) * 2 + H_) * 8) + np.minimum(N_, 7)) * 4 + i], 0.0)
                o = mfma(a, ha[nb][:, 4 * rq + i], o)
    got = np.zeros((32, 8))
    for l in range(64):
        for q in range(4):
            got[N_[l], 4 * H_[l] + q] = o[l, q]
    tsd = orc.load_weights(WEIGHTS_FP32)
    ref = orc.pointnet_encoder(tsd, torch.from_numpy(x.T[None]).float())[0].T.numpy()
    assert np.abs(got - ref).max() < 2e-5


def test_sdf_mlp_pack_matches_direct_mlp():
    sd = W.load_npz(WEIGHTS_FP32)
    pack = W.pack_sdf_mlp(sd).astype(np.float64)
    SD_W0, SD_W1 = 0, 6144
    SD_W2, SD_W3 = SD_W1 + 65536, SD_W1 + 2 * 65536
    SD_B0 = SD_W3 + 65536
    SD_WA, SD_BA = SD_B0 + 1024, SD_B0 + 1024 + 256
    assert pack.size == SD_BA + 4 + 2 * (409600 // 2)      # 32x32x16 split pack + 16x16x32 split pack
    DM = 128
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, size=(DM, 17))
    hl = np.zeros((32, 2, DM, 4))
    for f in range(17):
        hl[f >> 3, (f >> 2) & 1, :, f & 3] = x[:, f]

    def frag(off, w):
        v = np.zeros((64, 16))
        for q in range(4):
            for i in range(4):
                v[:, 4 * q + i] = pack[off + w * 32 + 8 * q + 4 * H_ + i]
        return v

    def layer(woff, boff, nkb, hl_in):
        accs = []
        for w in range(8):
            acc = [frag(boff, w) for _ in range(4)]
            for kb in range(nkb):
                for i in range(4):
                    a = pack[woff + w * nkb * 256 + kb * 256 + LANE * 4 + i]
                    for pt in range(4):
                        acc[pt] = mfma(a, hl_in[kb, H_, pt * 32 + N_, i], acc[pt])
            accs.append(acc)
        return accs

    def store(accs):
        out = np.zeros((32, 2, DM, 4))
        for w in range(8):
            for pt in range(4):
                for q in range(4):
                    for i in range(4):
                        out[4 * w + q, H_, pt * 32 + N_, i] = np.maximum(accs[w][pt][:, 4 * q + i], 0)
        return out

    hl = store(layer(SD_W0, SD_B0, 3, hl))
    hl = store(layer(SD_W1, SD_B0 + 256, 32, hl))
    hl = store(layer(SD_W2, SD_B0 + 512, 32, hl))
    accs = layer(SD_W3, SD_B0 + 768, 32, hl)
    alpha = np.full(DM, pack[SD_BA])
    for w in range(8):
        wa = frag(SD_WA, w)
        for pt in range(4):
            s = (wa * np.maximum(accs[w][pt], 0)).sum(1)  # per lane
            for l in range(64):
                alpha[pt * 32 + N_[l]] += s[l]
    tsd = orc.load_weights(WEIGHTS_FP32)
    ref = orc.geo_forward(tsd, torch.from_numpy(x).float())[:, 0].numpy()
    assert np.abs(alpha - ref).max() < 2e-5


# ---------------------------------------------------------------------------------------------
# split-operand (f16 hi + lo) layouts on v_mfma_f32_32x32x16_f16.  Operand slot jj of lane (n, h)
# pairs with slot jj of lane (m, h) of the other operand (verified on hardware by
# tools/probe_mfma_f16.hip); D layout as above.
# ---------------------------------------------------------------------------------------------
def mfma16(a, b, c):
    """a, b: [64, 8] per-lane operand slots; c: [64, 16]."""
    A = np.zeros((32, 2, 8))
    B = np.zeros((32, 2, 8))
    A[N_, H_] = a
    B[N_, H_] = b
    D = np.einsum("ihj,nhj->in", A, B)
    return c + D[ROW, N_[:, None]]


def _split(x):
    hi = np.asarray(x, np.float32).astype(np.float16)
    lo = (np.asarray(x, np.float32) - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def _mfma3(ah, al, bh, bl, c):
    return mfma16(ah, bh, mfma16(ah, bl, mfma16(al, bh, c)))


def test_sdf_mlp_split_pack_matches_direct_mlp():
    sd = W.load_npz(WEIGHTS_FP32)
    pack = W.pack_sdf_mlp(sd)
    SD_TOTAL = 6144 + 3 * 65536 + 1024 + 256 + 4
    fp = pack[:SD_TOTAL].astype(np.float64)
    halves = pack[SD_TOTAL:].view(np.float16).astype(np.float64)
    SD_B0 = 6144 + 3 * 65536
    SD_WA, SD_BA = SD_B0 + 1024, SD_B0 + 1280
    SH = [0, 16384, 16384 + 131072, 16384 + 2 * 131072]
    DM = 128
    J8 = np.arange(8)
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, size=(DM, 17)).astype(np.float32)
    xin = np.zeros((DM, 32), np.float32)
    xin[:, :17] = x

    def feature(ks, h, jj):
        return 16 * ks + 8 * (jj >> 2) + 4 * h + (jj & 3)

    def stage(vals, nks):     # vals [DM, 16 nks] -> hi/lo planes [nks][2][DM][8]
        hi = np.zeros((nks, 2, DM, 8))
        lo = np.zeros((nks, 2, DM, 8))
        for ks in range(nks):
            for h in range(2):
                a, b = _split(vals[:, feature(ks, h, J8)])
                hi[ks, h], lo[ks, h] = a, b
        return hi, lo

    def frag(off, w):
        v = np.zeros((64, 16))
        for q in range(4):
            for i in range(4):
                v[:, 4 * q + i] = fp[off + w * 32 + 8 * q + 4 * H_ + i]
        return v

    def layer(woff, boff, nks, hi, lo):
        accs = []
        for w in range(8):
            acc = [frag(boff, w) for _ in range(4)]
            for ks in range(nks):
                wi = woff + w * nks * 2 * 512 + ((ks * 2) * 64 + LANE[:, None]) * 8 + J8[None, :]
                ah, al = halves[wi], halves[wi + 512]
                for pt in range(4):
                    acc[pt] = _mfma3(ah, al, hi[ks, H_, pt * 32 + N_], lo[ks, H_, pt * 32 + N_], acc[pt])
            accs.append(acc)
        return accs

    def store(accs):
        hi = np.zeros((16, 2, DM, 8))
        lo = np.zeros((16, 2, DM, 8))
        for w in range(8):
            for pt in range(4):
                for ksl in range(2):
                    a, b = _split(np.maximum(accs[w][pt][:, 8 * ksl: 8 * ksl + 8], 0).astype(np.float32))
                    hi[2 * w + ksl, H_, pt * 32 + N_] = a
                    lo[2 * w + ksl, H_, pt * 32 + N_] = b
        return hi, lo

    hi, lo = stage(xin, 2)
    hi, lo = store(layer(SH[0], SD_B0, 2, hi, lo))
    hi, lo = store(layer(SH[1], SD_B0 + 256, 16, hi, lo))
    hi, lo = store(layer(SH[2], SD_B0 + 512, 16, hi, lo))
    accs = layer(SH[3], SD_B0 + 768, 16, hi, lo)
    alpha = np.full(DM, fp[SD_BA])
    for w in range(8):
        wa = frag(SD_WA, w)
        for pt in range(4):
            s = (wa * np.maximum(accs[w][pt], 0)).sum(1)
            for l in range(64):
                alpha[pt * 32 + N_[l]] += s[l]
    tsd = orc.load_weights(WEIGHTS_FP32)
    ref = orc.geo_forward(tsd, torch.from_numpy(x).float())[:, 0].numpy()
    assert np.abs(alpha - ref).max() < 2e-5


# ---------------------------------------------------------------------------------------------
# v_mfma_f32_16x16x32_f16 (k_lattice_table_x): A lane l holds A[m = l & 15][k = 8 (l >> 4) + jj], B lane l holds
# B[k = 8 (l >> 4) + jj][n = l & 15], D lane l register i holds D[m = 4 (l >> 4) + i][n = l & 15]
# (cdna_hip_programming.md section 3).
# ---------------------------------------------------------------------------------------------
M16, G16 = LANE & 15, LANE >> 4


def mfma16x16x32(a, b, c):
    """a, b: [64, 8] per-lane operand slots; c: [64, 4]."""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for jj in range(8):
        A[M16, 8 * G16 + jj] = a[:, jj]
        B[8 * G16 + jj, M16] = b[:, jj]
    D = A @ B
    return c + np.stack([D[4 * G16 + i, M16] for i in range(4)], 1)


def test_sdf_mlp_x_pack_matches_direct_mlp():
    """The 16x16x32 pack (weights._pack_split16) against the index arithmetic of k_lattice_table_x: octet layout of
    the activation planes, unit order of the weight fragments, bias / fc_alpha fragments, the 16 partials."""
    sd = W.load_npz(WEIGHTS_FP32)
    pack = W.pack_sdf_mlp(sd)
    SD_TOTAL = 6144 + 3 * 65536 + 1024 + 256 + 4
    assert pack.size == SD_TOTAL + 2 * 409600 // 2
    fp = pack[:SD_TOTAL].astype(np.float64)
    hx = pack[SD_TOTAL + 409600 // 2:].view(np.float16).astype(np.float64)
    SD_B0 = 6144 + 3 * 65536
    SD_WA, SD_BA = SD_B0 + 1024, SD_B0 + 1280
    SX = [0, 16384, 16384 + 131072, 16384 + 2 * 131072]
    DM = 128
    J8 = np.arange(8)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, size=(DM, 17)).astype(np.float32)
    xin = np.zeros((DM, 32), np.float32)
    xin[:, :17] = x
    # PARK: octet o of evaluation e = inputs 8 o .. 8 o + 7
    hi = np.zeros((4, DM, 8))
    lo = np.zeros((4, DM, 8))
    for o in range(4):
        hi[o], lo[o] = _split(xin[:, 8 * o: 8 * o + 8])

    def layer(woff, boff, n_units, hi, lo):
        accs = []
        for w in range(8):
            acc = [[np.stack([fp[boff + 32 * w + 16 * rb + 4 * G16 + i] for i in range(4)], 1) for _ in range(8)]
                   for rb in range(2)]
            for u in range(n_units):
                s, rb = u >> 1, u & 1
                wi = woff + (w * n_units + u) * 1024 + LANE[:, None] * 8 + J8[None, :]
                ah, al = hx[wi], hx[wi + 512]
                for cb in range(8):
                    bh, bl = hi[4 * s + G16, 16 * cb + M16], lo[4 * s + G16, 16 * cb + M16]
                    c = acc[rb][cb]
                    c = mfma16x16x32(ah, bl, c)
                    c = mfma16x16x32(al, bh, c)
                    acc[rb][cb] = mfma16x16x32(ah, bh, c)
            accs.append(acc)
        return accs

    def store(accs):
        hi = np.zeros((32, DM, 8))
        lo = np.zeros((32, DM, 8))
        for w in range(8):
            for cb in range(8):
                v = np.concatenate([accs[w][0][cb], accs[w][1][cb]], 1)      # [64 lanes][8]: one octet per lane
                a, b = _split(np.maximum(v, 0).astype(np.float32))
                hi[4 * w + G16, 16 * cb + M16] = a
                lo[4 * w + G16, 16 * cb + M16] = b
        return hi, lo

    hi, lo = store(layer(SX[0], SD_B0, 2, hi, lo))
    hi, lo = store(layer(SX[1], SD_B0 + 256, 16, hi, lo))
    hi, lo = store(layer(SX[2], SD_B0 + 512, 16, hi, lo))
    accs = layer(SX[3], SD_B0 + 768, 16, hi, lo)
    part = np.zeros((16, DM))
    for w in range(8):
        wa = [np.stack([fp[SD_WA + 32 * w + 16 * rb + 4 * G16 + i] for i in range(4)], 1) for rb in range(2)]
        for cb in range(8):
            s = sum((wa[rb] * np.maximum(accs[w][rb][cb], 0)).sum(1) for rb in range(2))     # [64]
            s = s + s[LANE ^ 32]
            for l in range(32):                                                           # lanes with g < 2
                part[w * 2 + G16[l], cb * 16 + M16[l]] = s[l]
    alpha = fp[SD_BA] + part.sum(0)
    tsd = orc.load_weights(WEIGHTS_FP32)
    ref = orc.geo_forward(tsd, torch.from_numpy(x).float())[:, 0].numpy()
    assert np.abs(alpha - ref).max() < 2e-5


def test_pointnet_x_pack_matches_direct_mlp():
    """The 16x16x32 point-encoder pack (weights._pack_pointnet_split16) against the index arithmetic of
    k_pointnet_scatter_x: inputs in K-group 0 of layer 1, register-to-operand chaining between the layers (row blocks
    2 s, 2 s + 1 -> K-step s), 8 output rows of the last layer."""
    sd = W.load_npz(WEIGHTS_FP32)
    pack = W.pack_pointnet(sd)
    PX_OFF = 34952
    assert pack.size == PX_OFF + 77824 // 2 + 4            # + certified-range trailer
    hx = pack[PX_OFF: PX_OFF + 77824 // 2].view(np.float16).astype(np.float64)
    fp = pack[:34952].astype(np.float64)
    PX_W1, PX_W2 = 0, 8192
    PX_W3, PX_W4 = PX_W2 + 32768, PX_W2 + 65536
    PN_B1 = 33536 + 1024
    rng = np.random.default_rng(4)
    x = rng.uniform(-1, 1, size=(32, 6)).astype(np.float32)      # 32 pairs: pair p = 16 cb + n
    J8 = np.arange(8)

    def bias4(layer, rb):
        return np.stack([fp[PN_B1 + layer * 128 + 16 * rb + 4 * G16 + i] for i in range(4)], 1)

    def frag(off):
        w = off + LANE[:, None] * 8 + J8[None, :]
        return hx[w], hx[w + 512]

    # layer 1: B fragments live in K-group 0 (lanes g = 0): slots 0..5 = the pair's six inputs
    bfr = []
    for cb in range(2):
        xin = np.zeros((64, 8), np.float32)
        xin[G16 == 0, :6] = x[16 * cb + M16[G16 == 0]]
        bfr.append(_split(xin))
    acc = [[None, None] for _ in range(8)]
    for rb in range(8):
        ah, al = frag(PX_W1 + rb * 1024)
        for cb in range(2):
            c = bias4(0, rb)
            c = mfma16x16x32(al, bfr[cb][0], c)
            c = mfma16x16x32(ah, bfr[cb][1], c)
            acc[rb][cb] = mfma16x16x32(ah, bfr[cb][0], c)

    def to_ops(acc):
        return [[_split(np.maximum(np.concatenate([acc[2 * s][cb], acc[2 * s + 1][cb]], 1), 0).astype(np.float32))
                 for cb in range(2)] for s in range(4)]

    def layer(woff, layer_idx, ops):
        out = [[bias4(layer_idx, rb) for _ in range(2)] for rb in range(8)]
        for s in range(4):
            for rb in range(8):
                ah, al = frag(woff + (s * 8 + rb) * 1024)
                for cb in range(2):
                    c = out[rb][cb]
                    c = mfma16x16x32(al, ops[s][cb][0], c)
                    c = mfma16x16x32(ah, ops[s][cb][1], c)
                    out[rb][cb] = mfma16x16x32(ah, ops[s][cb][0], c)
        return out

    acc = layer(PX_W2, 1, to_ops(acc))
    acc = layer(PX_W3, 2, to_ops(acc))
    ops = to_ops(acc)
    got = np.zeros((32, 8))
    for cb in range(2):
        o = np.stack([np.where(G16 < 2, fp[PN_B1 + 384 + np.minimum(4 * G16 + i, 7)], 0.0) for i in range(4)], 1)
        for s in range(4):
            ah, al = frag(PX_W4 + s * 1024)
            o = mfma16x16x32(al, ops[s][cb][0], o)
            o = mfma16x16x32(ah, ops[s][cb][1], o)
            o = mfma16x16x32(ah, ops[s][cb][0], o)
        for l in range(32):                                  # lanes g = 0, 1 hold output features 4 g .. 4 g + 3
            got[16 * cb + M16[l], 4 * G16[l]: 4 * G16[l] + 4] = o[l]
    tsd = orc.load_weights(WEIGHTS_FP32)
    ref = orc.pointnet_encoder(tsd, torch.from_numpy(x.T[None]).float())[0].T.numpy()
    assert np.abs(got - ref).max() < 2e-5


# ---------------------------------------------------------------------------------------------
# tcnn (FullyFusedMLP) layouts: f16 operands only, activations rounded to f16 between layers
# ---------------------------------------------------------------------------------------------
def _emulate_tcnn(halves, first_nks, x_padded, n_out_rows):
    """x_padded [32 evals, 16 * first_nks] -> network outputs [32, n_out_rows] with the kernels' index math."""
    J8 = np.arange(8)
    sf = 8 * (J8 >> 2)[None, :] + 4 * H_[:, None] + (J8 & 3)[None, :]        # slot -> feature within a K-step
    f16 = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float64)
    off = 0
    acc = [np.zeros((64, 16)) for _ in range(2)]
    for mb in range(2):
        for ks in range(first_nks):
            w = halves[off + ((mb * first_nks + ks) * 64 + LANE[:, None]) * 8 + J8[None, :]]
            b = f16(x_padded[N_[:, None], 16 * ks + sf])
            acc[mb] = mfma16(w, b, acc[mb])
    off += 2 * first_nks * 512
    for _ in range(2):
        s = [f16(np.maximum(acc[nb][:, 8 * ksl: 8 * ksl + 8], 0)) for nb in range(2) for ksl in range(2)]
        new = [np.zeros((64, 16)) for _ in range(2)]
        for mb in range(2):
            for g in range(4):
                w = halves[off + ((mb * 4 + g) * 64 + LANE[:, None]) * 8 + J8[None, :]]
                new[mb] = mfma16(w, s[g], new[mb])
        acc = new
        off += 2 * 4 * 512
    s = [f16(np.maximum(acc[nb][:, 8 * ksl: 8 * ksl + 8], 0)) for nb in range(2) for ksl in range(2)]
    o = np.zeros((64, 16))
    for g in range(4):
        o = mfma16(halves[off + (g * 64 + LANE[:, None]) * 8 + J8[None, :]], s[g], o)
    out = np.zeros((32, 32))
    out[ROW, N_[:, None]] = 0
    for l in range(64):
        for r in range(16):
            out[N_[l], ROW[l, r]] = o[l, r]
    return f16(out[:, :n_out_rows])


def test_tcnn_packs_match_oracle_restatement():
    from conftest import WEIGHTS_TCNN
    z = np.load(WEIGHTS_TCNN)
    rng = np.random.default_rng(4)
    # point encoder: 6 inputs padded to 16 with 1.0 -> first 8 outputs
    x = rng.uniform(-1, 1, size=(32, 6)).astype(np.float32)
    xp = np.ones((32, 16), np.float32)
    xp[:, :6] = x
    halves = W.pack_pointnet_tcnn(z["pointnet_backbone.model.params"]).view(np.float16).astype(np.float64)
    got = _emulate_tcnn(halves, 1, xp, 8)
    ref = orc.tcnn_mlp(torch.from_numpy(z["pointnet_backbone.model.params"]), torch.from_numpy(x), 16, 8).numpy()
    assert np.abs(got - ref).max() < 4e-3          # fp16 activations; accumulation order differs
    # SDF decoder: 17 inputs padded to 32 -> output 0
    x = rng.uniform(-1, 1, size=(32, 17)).astype(np.float32)
    xp = np.ones((32, 32), np.float32)
    xp[:, :17] = x
    halves = W.pack_sdf_tcnn(z["nerf.model.params"]).view(np.float16).astype(np.float64)
    got = _emulate_tcnn(halves, 2, xp, 1)
    ref = orc.tcnn_mlp(torch.from_numpy(z["nerf.model.params"]), torch.from_numpy(x), 32, 1).numpy()
    assert np.abs(got - ref).max() < 4e-3


def test_split_f16_precision_over_a_log_sweep():
    """The hi / lo split of the split-operand MLP modes (csrc/bnv_common.hpp: split8_f16; hi = rn16(x), lo = rn16(x - hi))
    against float64 on a log-spaced sweep, restated in numpy (the device form is bit-identical to this C++ form,
    tools/probe_split_mix.hip).  What the split carries: x - (hi + lo) is bounded by 2^-22 |x| while lo is a NORMAL
    f16 number, i.e. for |x| >= 2^-3 -- 22 significant bits; below that lo = x - hi falls under 2^-14 and is an
    f16 SUBNORMAL (kept by gfx950's f16 MFMA), whose spacing is 2^-24 whatever its size: the error is then absolute,
    <= 2^-25, so a weight of 0.05 is carried to ~20.6 bits and one of 0.001 to ~15.  Values below 2^-25 in magnitude
    can lose everything but hi.  (DESIGN.md section 3.3 states exactly this.)"""
    x = np.concatenate([np.logspace(-8, np.log10(6.0e4), 20001), -np.logspace(-8, np.log10(6.0e4), 20001)])
    x = x.astype(np.float32).astype(np.float64)               # the operands are fp32 values
    hi = x.astype(np.float32).astype(np.float16)
    lo = (x.astype(np.float32) - hi.astype(np.float32)).astype(np.float16)        # x - hi is exact in fp32
    assert np.isfinite(hi).all() and np.isfinite(lo).all()
    err = np.abs(x - (hi.astype(np.float64) + lo.astype(np.float64)))
    big = np.abs(x) >= 2.0 ** -3
    assert (err[big] <= 2.0 ** -22 * np.abs(x[big])).all()                   # lo normal: 11 + 11 significant bits
    assert (err[~big] <= 2.0 ** -25).all()                                   # lo subnormal: absolute, half its spacing
    # the bound is attained in order of magnitude on both sides (this is not a loose statement)
    assert (err[big] / np.abs(x[big])).max() > 2.0 ** -25 and err[~big].max() > 2.0 ** -27
    # significant bits carried at typical magnitudes
    bits = lambda v: float(-np.log2(err[np.isclose(np.abs(x), v, rtol=0.02)].max() / v))   # noqa: E731
    assert bits(1.0) >= 22.0 and 20.0 <= bits(0.05) <= 21.5 and 14.5 <= bits(0.001) <= 16.0
    # the f16 range guard of the modes is about hi: beyond 65,504 it overflows (tests/test_gpu_range_guard.py)
    assert np.isinf(np.float32(7.0e4).astype(np.float16))
