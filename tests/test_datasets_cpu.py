"""Sequence directory layout + 16-bit PNG codec (bnv_fusion_amd/datasets.py).  CPU only: the filter reversal is a
host function of the shared library."""
import os

import numpy as np
import pytest

from bnv_fusion_amd import datasets


@pytest.mark.parametrize("filter_type", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(1, 1), (7, 5), (48, 64)])
def test_png16_round_trip_every_filter(tmp_path, filter_type, shape):
    rng = np.random.default_rng(filter_type * 10 + shape[0])
    img = rng.integers(0, 65536, size=shape, dtype=np.uint16)
    if shape[0] > 1:
        img[1] = img[0]                                 # identical rows / smooth runs exercise the predictors
        img[:, : shape[1] // 2] = np.sort(img[:, : shape[1] // 2], axis=1)
    p = datasets.write_png16(str(tmp_path / "d.png"), img, filter_type)
    back = datasets.read_png16(p)
    assert back.dtype == np.uint16 and np.array_equal(back, img)


def test_png_rejects_what_it_does_not_decode(tmp_path):
    p = tmp_path / "x.png"
    p.write_bytes(b"not a png")
    with pytest.raises(ValueError):
        datasets.read_png16(str(p))


def test_sequence_layout_round_trip(tmp_path):
    from bnv_fusion_amd import synthetic
    depths = [synthetic.depth_u16(t, 60, 80) for t in range(5)]
    poses = [synthetic.pose(t) for t in range(5)]
    K = synthetic.intrinsics(60, 80)
    root = datasets.write_sequence(str(tmp_path), "scene3d/demo", depths, K, poses, [2.52, 2.52, 2.52])
    assert sorted(os.listdir(root)) == ["depth", "image", "pose"]
    ds = datasets.FusionInferenceDataset(str(tmp_path), "scene3d/demo", skip_images=2, device="cpu")
    assert len(ds) == 3 and np.allclose(ds.dimensions, 2.52)
    for k, fr in enumerate(ds):
        i = 2 * k
        assert fr["frame_id"] == i and np.array_equal(fr["depth"].numpy(), depths[i])
        assert np.array_equal(fr["T_wc"], poses[i].astype(np.float32).astype(np.float64))     # read_pose is float32
        assert np.array_equal(fr["intr_mat"], K.astype(np.float32).astype(np.float64))
    half = datasets.FusionInferenceDataset(str(tmp_path), "scene3d/demo", downsample_scale=0.5, device="cpu")[0]
    assert tuple(half["depth"].shape) == (30, 40) and np.isclose(half["intr_mat"][0, 0], K[0, 0] * 0.5)
    assert np.array_equal(half["depth"].numpy(), depths[0][::2, ::2])


def test_random_subset_is_a_sample_without_replacement_in_random_order():
    """optimize.random_subset stands in for ``torch.randperm(n)[:k]`` (fusion_inference_dataset.py:383) without permuting
    all n: k distinct indices, every index equally likely, every index equally likely to come first; large k falls back to
    the permutation."""
    import torch
    from bnv_fusion_amd.optimize import random_subset
    g = torch.Generator().manual_seed(3)
    x = random_subset(640 * 480, 5000, "cpu", g)
    assert x.shape == (5000,) and x.dtype == torch.int64 and x.unique().numel() == 5000
    assert 0 <= int(x.min()) and int(x.max()) < 640 * 480
    assert (x[1:] < x[:-1]).float().mean() > 0.4          # (not sorted: draw order)
    n, k, reps = 64, 8, 8000
    cnt, first = torch.zeros(n), torch.zeros(n)
    for _ in range(reps):
        y = random_subset(n, k, "cpu", g)
        assert y.unique().numel() == k
        cnt[y] += 1
        first[y[0]] += 1
    exp = reps * k / n
    assert float((cnt - exp).abs().max()) < 6 * (exp * (1 - k / n)) ** 0.5
    assert float((first - reps / n).abs().max()) < 6 * (reps / n) ** 0.5
    z = random_subset(40, 8, "cpu", g)                    # k > n / 8: the permutation itself
    assert z.unique().numel() == 8 and int(z.max()) < 40

