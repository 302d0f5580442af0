"""Capture golden input/output vectors from the reference itself.

Build-container only (needs /root/reference):  python tests/golden/make_golden.py
The reference is imported under tests/golden/ref_shims.py and run on CPU with the
fp32 checkpoint; only DATA (inputs + the reference's outputs) is written, as
tests/golden/*.npz.  Cases:

  encode_64.npz     one frame, 64^3 grid (dims 1.24 m, voxel 0.02): encode_pointcloud
                    sparse + dense outputs; includes out-of-bounds points, points inside the
                    1-voxel margin and points on exact-integer voxel coordinates (floor==ceil).
  sequence_64.npz   12 frames fused with encode_pointcloud + _integrate -> final volume.
  decode_64.npz     SparseVolume.decode_pts on that volume: 3x3x3 lattice of 150 voxels and
                    400 random points, query_tensor True/False, with/without sdf_delta,
                    is_coords True/False; plus count_optim.
  dense_decode_64.npz  decode_feature_grid_w_pts on dense grids of one frame.
  encode_128.npz    one 160x120 synthetic depth frame, 128^3 grid (dims 2.52, voxel 0.02).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_shims  # noqa: E402


def surface_points(n, seed, voxel, extent, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    xy = (torch.rand(n, 2, generator=g) - 0.5) * 2 * extent
    z = 0.12 * torch.sin(xy[:, 0] * 6 + shift) * torch.cos(xy[:, 1] * 5) + 0.003 * torch.randn(n, generator=g)
    xyz = torch.cat([xy, z[:, None]], -1)
    nrm = torch.stack([-0.72 * torch.cos(xy[:, 0] * 6 + shift) * torch.cos(xy[:, 1] * 5),
                       0.6 * torch.sin(xy[:, 0] * 6 + shift) * torch.sin(xy[:, 1] * 5),
                       torch.ones(n)], -1)
    nrm = torch.nn.functional.normalize(nrm + 0.05 * torch.randn(n, 3, generator=g), dim=-1)
    return torch.cat([xyz, nrm], -1).float()


def edge_case_points(vol, voxel):
    """Points that exercise the bounds mask and the floor==ceil duplicates."""
    bmin, bmax = vol.min_coords, vol.max_coords
    pts = []
    # exact-integer voxel coordinates on 1, 2 and 3 axes, repeated so some voxels pass min_pts
    for rep in range(12):
        base = bmin + voxel * torch.tensor([20.0, 21.0, 22.0])
        pts.append(base)                                             # integer on x, y, z
        pts.append(base + torch.tensor([0.3 * voxel, 0.0, 0.0]))      # integer on y, z
        pts.append(base + torch.tensor([0.3 * voxel, 0.6 * voxel, 0.0]))  # integer on z
    # outside / inside the one-voxel margin, both sides
    for ax in range(3):
        for d in (-0.5, 0.5, 1.0, 1.5):
            p = torch.zeros(3)
            p[ax] = bmin[ax] + d * voxel
            pts.append(p.clone())
            p[ax] = bmax[ax] - d * voxel
            pts.append(p.clone())
    pts.append(torch.tensor([10.0, 0.0, 0.0]))
    pts.append(torch.tensor([0.0, -10.0, 0.0]))
    xyz = torch.stack(pts)
    nrm = torch.nn.functional.normalize(torch.ones_like(xyz) * torch.tensor([0.2, -0.3, 0.9]), dim=-1)
    return torch.cat([xyz, nrm], -1).float()


def t2n(t):
    return None if t is None else t.detach().cpu().numpy()


def main():
    torch.set_num_threads(8)
    voxel = 0.02
    dims = np.array([1.24, 1.24, 1.24])
    model, SV = ref_shims.build_reference_model(voxel, "/tmp/refwork")

    # ---------------- encode, 64^3 ------------------------------------------------------------
    vol = SV(8, voxel, dims, 8, device="cpu")
    assert vol.n_xyz.tolist() == [64, 64, 64]
    pts = torch.cat([surface_points(16000, 1, voxel, 0.55), edge_case_points(vol, voxel)], 0)[None]
    with torch.no_grad():
        feats, pcounts, flat_ids, grid_ids, n_avg = model.encode_pointcloud(
            pts.clone(), vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size, return_dense=False)
        fg, mask, uids, flat_all = model.encode_pointcloud(
            pts.clone(), vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size, return_dense=True)
    nz = mask[0, 0].reshape(-1).nonzero()[:, 0]
    np.savez_compressed(
        os.path.join(HERE, "encode_64.npz"),
        input_pts=t2n(pts), dims=dims, voxel_size=voxel, n_xyz=t2n(vol.n_xyz),
        min_coords=t2n(vol.min_coords), max_coords=t2n(vol.max_coords),
        feats=t2n(feats), pcounts=t2n(pcounts), flat_ids=t2n(flat_ids), grid_ids=t2n(grid_ids),
        n_avg_pts=t2n(n_avg),
        dense_unique_flat_ids=t2n(uids), dense_counts=t2n(mask[0, 0].reshape(-1)[uids]),
        dense_feats=t2n(fg[0].reshape(8, -1)[:, uids].T), dense_nonzero=t2n(nz),
        dense_flat_ids_all=t2n(flat_all[0].to(torch.int32)))
    print("encode_64:", feats.shape, float(n_avg), "U =", len(uids))

    # empty -> 5 x None
    far = pts.clone()
    far[..., :3] += 100.0
    assert model.encode_pointcloud(far, vol.n_xyz, vol.min_coords, vol.max_coords, voxel, False)[0] is None

    # ---------------- 12-frame sequence -> volume ----------------------------------------------
    vol = SV(8, voxel, dims, 8, device="cpu")
    frames = []
    for t in range(12):
        p = surface_points(6000, 100 + t, voxel, 0.2, shift=0.02 * t)[None]
        frames.append(t2n(p))
        with torch.no_grad():
            f, c, _, g, n = model.encode_pointcloud(p.clone(), vol.n_xyz, vol.min_coords, vol.max_coords,
                                                    vol.voxel_size, return_dense=False)
            vol.track_n_pts(n)
            model._integrate(vol, g, f, c)
    vol.to_tensor()
    keys = vol.active_coordinates
    order = np.lexsort((keys[:, 2].numpy(), keys[:, 1].numpy(), keys[:, 0].numpy()))
    np.savez_compressed(
        os.path.join(HERE, "sequence_64.npz"),
        frames=np.stack(frames), dims=dims, voxel_size=voxel,
        keys_sorted=t2n(keys)[order], features_sorted=t2n(vol.features)[order],
        weights_sorted=t2n(vol.weights)[order], num_hits_sorted=t2n(vol.num_hits)[order],
        keys_insertion=t2n(keys), n_pts_list=np.asarray(vol.n_pts_list))
    print("sequence_64: M =", len(keys), "weights>=8:", int((vol.weights >= 8).sum()))

    # ---------------- decode_pts ----------------------------------------------------------------
    g = torch.Generator().manual_seed(7)
    valid_rows = (vol.weights[:, 0] >= 8).nonzero()[:, 0]
    sel = valid_rows[torch.randperm(len(valid_rows), generator=g)[:150]]
    origins = keys[sel]
    r = np.arange(0, 1.5, 0.5) - 0.5
    lat = np.stack(np.meshgrid(r, r, r, indexing="ij"), -1)
    vc = torch.from_numpy((np.tile(lat, (len(origins), 1, 1, 1, 1))
                           + origins.numpy()[:, None, None, None, :]).reshape(1, len(origins), 27, 3)).float()
    rnd = keys[valid_rows[torch.randint(len(valid_rows), (400,), generator=g)]].float() \
        + (torch.rand(400, 3, generator=g) - 0.5) * 1.6
    rnd[:40] = torch.round(rnd[:40] * 2) / 2        # some exact integer / half coordinates
    rnd = rnd.reshape(1, 50, 8, 3)
    rnd_world = rnd * voxel + vol.min_coords
    # TSDF prior on the reference's 0.025 m grid (run_e2e.py:62-71,169-186); random values
    from src.utils import voxel_utils as ref_vu
    _, _, n_tsdf = ref_vu.get_world_range(dims, 0.025)
    sdf_delta = (torch.rand([1, 1] + list(n_tsdf), generator=g) - 0.5) * 0.02
    with torch.no_grad():
        out = dict(
            lattice_qt=vol.decode_pts(vc, model.nerf, None, is_coords=True, query_tensor=True),
            lattice_q=vol.decode_pts(vc, model.nerf, None, is_coords=True, query_tensor=False),
            lattice_delta=vol.decode_pts(vc, model.nerf, sdf_delta, is_coords=True, query_tensor=True),
            random_qt=vol.decode_pts(rnd, model.nerf, None, is_coords=True, query_tensor=True),
            random_world_out=vol.decode_pts(rnd_world, model.nerf, None, is_coords=False, query_tensor=False),
            random_delta=vol.decode_pts(rnd, model.nerf, sdf_delta, is_coords=True, query_tensor=True),
        )
        # count_optim on the neighbours of the random points, then decode again
        from src.models.fusion.utils import get_neighbors as ref_get_neighbors
        w_before = vol.weights.clone()
        vol.count_optim(ref_get_neighbors(rnd))
        out["weights_after_count_optim_sorted"] = vol.weights.clone()[order]
        out["random_after_count_optim"] = vol.decode_pts(rnd, model.nerf, None, is_coords=True, query_tensor=True)
        vol.weights.copy_(w_before)
    np.savez_compressed(
        os.path.join(HERE, "decode_64.npz"),
        origins=t2n(origins), lattice_coords=t2n(vc), random_coords=t2n(rnd),
        random_world_coords=t2n(rnd_world), sdf_delta=t2n(sdf_delta), **{k: t2n(v) for k, v in out.items()})
    print("decode_64: lattice valid frac",
          float((out["lattice_qt"] != voxel).float().mean()), "random valid frac",
          float((out["random_qt"] != voxel).float().mean()))

    # ---------------- dense decode -------------------------------------------------------------
    with torch.no_grad():
        p = torch.from_numpy(frames[0])
        p8 = torch.cat([p] * 1, 1)
        fg, mask, uids, _ = model.encode_pointcloud(p8.clone(), vol.n_xyz, vol.min_coords, vol.max_coords,
                                                    voxel, return_dense=True)
        u3 = ref_vu.unflatten(uids[mask[0, 0].reshape(-1)[uids] >= 8], vol.n_xyz).float()
        q = u3[torch.randint(len(u3), (600,), generator=g)] + (torch.rand(600, 3, generator=g) - 0.5) * 1.4
        q[:60] = torch.round(q[:60] * 2) / 2
        q[60:140] += (torch.rand(80, 3, generator=g) - 0.5) * 6   # some queries off the surface -> invalid
        q = q.clamp(0.0, 62.9)[None]
        sdf, _ = model.decode_feature_grid_w_pts(q, fg, mask, voxel, vol.min_coords, global_coords=False)
    np.savez_compressed(
        os.path.join(HERE, "dense_decode_64.npz"),
        input_pts=t2n(p8), queries=t2n(q), sdf=t2n(sdf), voxel_size=voxel, dims=dims)
    print("dense_decode_64: valid frac", float((sdf != voxel).float().mean()))

    # ---------------- encode, 128^3 from a synthetic depth frame -------------------------------
    from oracle import bnv_oracle as orc
    dims128 = np.array([2.52, 2.52, 2.52])
    vol = SV(8, voxel, dims128, 8, device="cpu")
    assert vol.n_xyz.tolist() == [128, 128, 128]
    depth = orc.synthetic_depth(3, H=120, W=160)
    intr = orc.SYNTHETIC_INTRINSICS.copy()
    intr[:2] *= 0.25
    p = torch.from_numpy(orc.depth_to_input_pts(depth, intr, orc.synthetic_pose(3))).float()[None]
    with torch.no_grad():
        feats, pcounts, flat_ids, grid_ids, n_avg = model.encode_pointcloud(
            p.clone(), vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size, return_dense=False)
    np.savez_compressed(
        os.path.join(HERE, "encode_128.npz"),
        input_pts=t2n(p), dims=dims128, voxel_size=voxel, feats=t2n(feats), pcounts=t2n(pcounts),
        flat_ids=t2n(flat_ids), grid_ids=t2n(grid_ids), n_avg_pts=t2n(n_avg))
    print("encode_128:", feats.shape, float(n_avg))


if __name__ == "__main__":
    main()
