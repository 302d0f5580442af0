"""Golden vectors for the global-optimiser edge, captured from the reference.

Build-container only (needs /root/reference):  python tests/golden/make_golden_grad.py
Re-creates the fused 64^3 volume of sequence_64.npz inside the reference's SparseVolume, makes
``volume.features`` an nn.Parameter as run_e2e.py:114 does and records

  decode_grad_64.npz   d(sum(out * g)) / d(features) through SparseVolume.decode_pts (query_tensor=True)
                       for the random and the lattice queries of decode_64.npz;
  optimize_64.npz      one calculate_loss (src/utils/render_utils.py:551-590) on synthetic rays with the
                       CPU RNG seeded: inputs, the sampled points, the loss, weights after count_optim and
                       d loss / d features.

Only DATA is written.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_shims  # noqa: E402


def t2n(t):
    return t.detach().cpu().numpy()


def surface_z(x, y, shift=0.0):
    return 0.12 * torch.sin(x * 6 + shift) * torch.cos(y * 5)


def make_rays(n, seed):
    """Camera 0.45 m above the surface patch of sequence_64 looking down -z; gt points lie on the analytic
    surface, uv are their (sub-pixel) projections, neighbours are 9 nearby surface points."""
    g = torch.Generator().manual_seed(seed)
    T_wc = torch.eye(4)
    T_wc[:3, :3] = torch.tensor([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])
    T_wc[:3, 3] = torch.tensor([0.01, -0.02, 0.45])
    intr = torch.tensor([[100.0, 0, 40.0], [0, 100.0, 30.0], [0, 0, 1.0]])
    xy = (torch.rand(n, 2, generator=g) - 0.5) * 0.34
    gt = torch.stack([xy[:, 0], xy[:, 1], surface_z(xy[:, 0], xy[:, 1])], -1)
    pc = (T_wc[:3, :3].T @ (gt - T_wc[:3, 3]).T).T
    uv = torch.stack([pc[:, 0] / pc[:, 2] * 100.0 + 40.0, pc[:, 1] / pc[:, 2] * 100.0 + 30.0], -1)
    off = (torch.rand(n, 9, 2, generator=g) - 0.5) * 0.012
    off[:, 4] = 0
    nxy = xy[:, None, :] + off
    nb = torch.stack([nxy[..., 0], nxy[..., 1], surface_z(nxy[..., 0], nxy[..., 1])], -1)
    nb_mask = (torch.rand(n, 9, generator=g) > 0.15).float()
    nb_mask[:, 4] = 1
    mask = (torch.rand(n, generator=g) > 0.1).float()
    return {
        "uv": uv[None].float(), "rgb": torch.zeros(1, n, 3), "gt_pts": gt[None].float(),
        "intr_mat": intr[None], "T_wc": T_wc[None], "mask": mask[None],
        "neighbor_pts": nb[None].float(), "neighbor_masks": nb_mask[None]}


def main():
    torch.set_num_threads(8)
    voxel = 0.02
    dims = np.array([1.24, 1.24, 1.24])
    model, SV = ref_shims.build_reference_model(voxel, "/tmp/refwork")
    seq = np.load(os.path.join(HERE, "sequence_64.npz"))
    dec = np.load(os.path.join(HERE, "decode_64.npz"))
    pos = {tuple(k): i for i, k in enumerate(seq["keys_sorted"].tolist())}
    perm = np.array([pos[tuple(k)] for k in seq["keys_insertion"].tolist()])
    vol = SV(8, voxel, dims, 8, device="cpu")
    vol.insert(torch.from_numpy(seq["keys_insertion"]), torch.from_numpy(seq["features_sorted"][perm]),
               torch.from_numpy(seq["weights_sorted"][perm]), torch.from_numpy(seq["num_hits_sorted"][perm]))
    vol.to_tensor()
    assert np.array_equal(t2n(vol.active_coordinates), seq["keys_insertion"])
    vol.features = torch.nn.Parameter(vol.features)
    g = torch.Generator().manual_seed(11)
    out = {}
    sdf_delta = torch.from_numpy(dec["sdf_delta"])
    for name, key in (("random", "random_coords"), ("lattice", "lattice_coords")):
        c = torch.from_numpy(dec[key])
        go = torch.randn(list(c.shape[:-1]) + [1], generator=g) * 1e-3
        vol.features.grad = None
        sdf = vol.decode_pts(c, model.nerf, sdf_delta, is_coords=True, query_tensor=True)
        (sdf * go).sum().backward()
        out[f"{name}_grad_out"] = t2n(go)
        out[f"{name}_sdf"] = t2n(sdf)
        out[f"{name}_grad_features"] = t2n(vol.features.grad)
        print(name, "grad rows touched:", int((vol.features.grad.abs().sum(-1) > 0).sum()), "of", len(perm),
              "max |g|", float(vol.features.grad.abs().max()))
    np.savez_compressed(os.path.join(HERE, "decode_grad_64.npz"), **out)

    # ---------------- one calculate_loss of the global optimiser ---------------------------------
    from src.utils.render_utils import calculate_loss
    import src.utils.render_utils as ru
    rays = make_rays(160, 21)
    truncated_units = 10
    truncated_dist = min(truncated_units * voxel * 0.5, 0.1)
    ray_max_dist = 3
    captured = {}
    orig = ru.hierarchical_sampling

    def spy(*a, **k):
        pts, dists = orig(*a, **k)
        captured["pts"], captured["dists"] = pts.clone(), dists.clone()
        return pts, dists

    ru.hierarchical_sampling = spy
    w_before = vol.weights.clone()
    vol.features.grad = None
    torch.manual_seed(1234)
    loss = calculate_loss(vol, rays, model.nerf, truncated_units=truncated_units, truncated_dist=truncated_dist,
                          ray_max_dist=ray_max_dist, sdf_delta=sdf_delta)
    ru.hierarchical_sampling = orig
    total = sum(v for k, v in loss.items() if k[0] != "_")
    total.backward()
    print("loss", {k: float(v) for k, v in loss.items()}, "pts", tuple(captured["pts"].shape),
          "grad rows", int((vol.features.grad.abs().sum(-1) > 0).sum()),
          "weights bumped", int((vol.weights != w_before).sum()))
    np.savez_compressed(
        os.path.join(HERE, "optimize_64.npz"),
        **{"rays_" + k: t2n(v) for k, v in rays.items()},
        seed=1234, truncated_units=truncated_units, truncated_dist=truncated_dist, ray_max_dist=ray_max_dist,
        pts=t2n(captured["pts"]), dists=t2n(captured["dists"]),
        depth_bce_loss=float(loss["depth_bce_loss"]), grad_features=t2n(vol.features.grad),
        weights_after=t2n(vol.weights), weights_before=t2n(w_before))


if __name__ == "__main__":
    main()
