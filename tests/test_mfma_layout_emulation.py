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
    assert pack.size == PN_B4 + 8 == 34952
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
    assert pack.size == SD_BA + 4
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
