"""Global optimisation of the fused feature volume against the depth frames -- the second level of
the reference's bi-level fusion (src/run_e2e.py:111-162, src/utils/render_utils.py:77-94, 191-233,
411-590, src/datasets/fusion_inference_dataset.py:308-420).

Function names and arguments mirror the reference module so a maintainer can swap
``from src.utils.render_utils import calculate_loss`` for this one.  Everything here is device-side
torch glue (a few thousand rays per step); the heavy part -- ``SparseVolume.decode_pts`` forward and
its backward into ``volume.features`` -- are the HIP kernels behind ``bnv_decode_pts`` /
``bnv_decode_pts_backward``.

Randomness: pass ``generator=`` (a CPU or device ``torch.Generator``) for reproducible draws; a CPU
generator reproduces the reference's CPU stream bit for bit (used by the parity tests).
"""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .fusion import get_neighbors


def _rand(shape, device, generator):
    if generator is not None and generator.device.type == "cpu":
        return torch.rand(*shape, generator=generator).to(device)
    return torch.rand(*shape, device=device, generator=generator)


def lift(x, y, z, intrinsics):
    """render_utils.py:411-428: pixel (x, y) at depth z -> homogeneous camera coordinates."""
    intrinsics = intrinsics.to(x.device)
    fx, fy = intrinsics[:, 0, 0].unsqueeze(-1), intrinsics[:, 1, 1].unsqueeze(-1)
    cx, cy = intrinsics[:, 0, 2].unsqueeze(-1), intrinsics[:, 1, 2].unsqueeze(-1)
    sk = intrinsics[:, 0, 1].unsqueeze(-1)
    x_lift = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_lift = (y - cy) / fy * z
    return torch.stack((x_lift, y_lift, z, torch.ones_like(z)), dim=-1)


def get_camera_params(uv, pose, intrinsics):
    """render_utils.py:431-458 for 4x4 camera-to-world poses -> (unit ray_dirs [b, n, 3], cam_loc [b, 3])."""
    if pose.shape[1] == 7:
        raise NotImplementedError("quaternion poses are not used on this path (run_e2e.py passes T_wc 4x4)")
    cam_loc = pose[:, :3, 3]
    z_cam = uv[:, :, 0] * 0.0 + 1.0
    cam_pts = lift(uv[:, :, 0], uv[:, :, 1], z_cam, intrinsics).permute(0, 2, 1)
    world = torch.bmm(pose, cam_pts).permute(0, 2, 1)[:, :, :3]
    return F.normalize(world - cam_loc[:, None, :], dim=2), cam_loc


def stratified_sampling(n_pts, n_samples, distances, generator=None):
    """render_utils.py:77-94.  distances [(b), N, 1] -> one uniform sample per stratum, [b, N, S, 1]."""
    if distances.dim() < 3:
        distances = distances.unsqueeze(0)
    b, n_pts = distances.shape[:2]
    edges = torch.linspace(0, 1, steps=n_samples, device=distances.device).unsqueeze(0).repeat(b, n_pts, 1)
    edges = edges * distances
    mids = 0.5 * (edges[..., 1:] + edges[..., :-1])
    upper = torch.cat([mids, edges[..., -1:]], dim=-1)
    lower = torch.cat([edges[..., :1], mids], dim=-1)
    t = _rand((b, n_pts, n_samples), distances.device, generator)
    return (lower + (upper - lower) * t).unsqueeze(-1)


def hierarchical_sampling(n_fine_samples, n_coarse_samples, depths, surface, ray_directions, cam_loc,
                          offset_distance=0.5, max_depth=5.0, generator=None):
    """render_utils.py:191-233: fine samples within +-offset_distance of the observed surface plus coarse
    samples from the camera to the surface, sorted along the ray -> (pts [b, N, S, 3], dists [b, N, S, 1])."""
    n_pts = ray_directions.shape[1]
    back = torch.where(depths - offset_distance < 0, depths, torch.zeros_like(depths) + offset_distance)
    start_pts = surface - back.unsqueeze(-1) * ray_directions
    start_depths = torch.sqrt(torch.sum((start_pts - cam_loc.unsqueeze(1)) ** 2, dim=-1))
    span = torch.zeros_like(ray_directions[:, :, :1]) + offset_distance * 2
    fine = stratified_sampling(n_pts, n_fine_samples, span, generator)
    fine = fine + start_depths.unsqueeze(-1).unsqueeze(-1)
    coarse = stratified_sampling(n_pts, n_coarse_samples, depths.unsqueeze(-1), generator)
    dists, _ = torch.sort(torch.cat([fine, coarse], -2), -2)
    pts = cam_loc.unsqueeze(1).unsqueeze(1) + dists * ray_directions.unsqueeze(2)
    return pts, dists


def render_with_rays(volume, rays, nerf, sdf_delta, truncated_units, truncated_dist, ray_max_dist,
                     generator=None):
    """render_utils.py:461-505: sample every ray, bump the optimisation counter of the touched voxels
    and decode the SDF at the samples (differentiable w.r.t. ``volume.features``)."""
    ray_dirs, cam_loc = get_camera_params(rays["uv"], rays["T_wc"], rays["intr_mat"])
    gt_depths = torch.sqrt(torch.sum((rays["gt_pts"] - cam_loc.unsqueeze(1)) ** 2, dim=-1))
    pts, dists = hierarchical_sampling(truncated_units * 2, int(ray_max_dist * 5), gt_depths, rays["gt_pts"],
                                       ray_dirs, cam_loc, offset_distance=truncated_dist,
                                       max_depth=ray_max_dist, generator=generator)
    coords = (pts - volume.min_coords) / volume.voxel_size
    volume.count_optim(get_neighbors(coords))
    pred_sdf = volume.decode_pts(pts, nerf, sdf_delta=sdf_delta)[..., 0]
    return {"cam_loc": cam_loc, "ray_dirs": ray_dirs, "sdf_on_rays": pred_sdf, "pts_on_rays": pts}


def compute_sdf_loss(rays, pred_sdf, pred_pts, cam_loc, num_valid_pixels, truncated_dist):
    """render_utils.py:508-549: L1 between the decoded SDF and the signed distance to the nearest valid
    surface point of the pixel's 3x3 neighbourhood, on samples in front of / just behind the surface."""
    gt_depths = torch.sqrt(torch.sum((rays["gt_pts"] - cam_loc.unsqueeze(1)) ** 2, dim=-1)).unsqueeze(-1)
    depths = torch.sqrt(torch.sum((pred_pts - cam_loc.unsqueeze(1).unsqueeze(1)) ** 2, dim=-1))
    gt_sdf = torch.clip(gt_depths - depths, min=-truncated_dist, max=truncated_dist)
    valid_map = gt_sdf > max(-truncated_dist * 0.5, -0.05)
    d = torch.sqrt(torch.sum((rays["neighbor_pts"].unsqueeze(2) - pred_pts.unsqueeze(3)) ** 2, dim=-1))
    nb_mask = rays["neighbor_masks"].unsqueeze(2).repeat(1, 1, pred_pts.shape[2], 1)
    d = torch.where(nb_mask.bool(), d, torch.ones_like(d) * 10000)
    nearest = torch.min(d, dim=-1)[0]
    sign = torch.where(gt_sdf > 0, torch.ones_like(gt_sdf), torch.ones_like(gt_sdf) * -1)
    target = torch.clip(nearest * sign, min=-truncated_dist, max=truncated_dist)
    l1 = F.l1_loss(pred_sdf, target, reduction="none") * valid_map
    return (l1 * rays["mask"].unsqueeze(-1)).sum() / num_valid_pixels


def calculate_loss(volume, rays, nerf, truncated_units, truncated_dist, ray_max_dist, sdf_delta=None,
                   generator=None):
    """render_utils.py:551-590 -> {"depth_bce_loss": scalar}."""
    num_valid_pixels = torch.sum(rays["mask"]) + 1e-4
    out = render_with_rays(volume, rays, nerf, sdf_delta, truncated_units, truncated_dist, ray_max_dist,
                           generator=generator)
    loss = compute_sdf_loss(rays, out["sdf_on_rays"], out["pts_on_rays"], out["cam_loc"], num_valid_pixels,
                            truncated_dist)
    return {"depth_bce_loss": loss}


def key_frame_points(depth, intr_mat, T_wc, ray_max_dist):
    """The part of _sample_key_frame that depends on the frame alone: every pixel's world point (float64 arithmetic of
    geometry.py:150-171, rounded to float32 as the sampler's ``.float()`` does after its gather) and validity
    (common.py:110-113).  -> (pts [H * W, 3] f32, mask [H * W] f32, H, W, host copies of T_wc / intr_mat)."""
    dev = depth.device
    intr_mat, T_wc = torch.as_tensor(intr_mat), torch.as_tensor(T_wc)
    depth = depth.to(torch.float64)
    mask = (depth > 0) & (depth < ray_max_dist)                       # common.py:110-113
    depth = depth * mask
    H, W = depth.shape
    K = intr_mat.to(dev, torch.float32)
    T = T_wc.to(dev, torch.float32).to(torch.float64)
    # geometry.py:163-168 forms the normalised pixel coordinates in float32 before the float64 product
    u = ((torch.arange(W, device=dev, dtype=torch.float32) - K[0, 2]) / K[0, 0]).to(torch.float64)
    v = ((torch.arange(H, device=dev, dtype=torch.float32) - K[1, 2]) / K[1, 1]).to(torch.float64)
    pts_c = torch.stack([u[None, :].expand(H, W), v[:, None].expand(H, W), torch.ones_like(depth)], -1)
    pts_c = pts_c * depth[..., None]                                  # geometry.py:150-171
    pts_w = pts_c.reshape(-1, 3) @ T[:3, :3].T + T[:3, 3]
    r = torch.arange(-1, 2, device=dev)
    oy, ox = torch.meshgrid(r, r, indexing="ij")                      # np.meshgrid(range_, range_) order: x fastest
    return {"pts": pts_w.float(), "mask": mask.reshape(-1).float(), "H": H, "W": W,
            "oy": oy.reshape(1, -1), "ox": ox.reshape(1, -1), "zero_rgb": {},
            "intr_mat": intr_mat.to(dev).float().reshape(1, 3, 3), "T_wc": T_wc.to(dev).float().reshape(1, 4, 4),
            "T_wc_host": T_wc.detach().cpu().numpy().astype(np.float32).reshape(4, 4),
            "intr_host": intr_mat.detach().cpu().numpy().astype(np.float32).reshape(3, 3)}


def random_subset(n, k, device, generator=None):
    """``torch.randperm(n)[:k]`` in distribution -- k indices out of n without replacement, in random order
    (fusion_inference_dataset.py:383) -- without permuting all n on the device: a full permutation of a 640x480 image is a
    sort of 307,200 keys (18 merge passes, 135 us per optimiser step) to pick 5,000 rays.  2 k indices are drawn WITH
    replacement and every repeat of an earlier draw is dropped, which is sequential sampling with rejection -- the same
    distribution; the first k survivors are kept, in draw order.  With k <= n / 8 the expected number of repeats among the
    2 k draws is <= k / 4 and fewer than k survivors would need more than k of them (never seen; such a slot would hold
    index 0).  Larger k: the permutation."""
    if k * 8 > n:
        return torch.randperm(n, device=device, generator=generator)[:k]
    m = 2 * k
    draws = torch.randint(n, (m,), device=device, generator=generator)
    s, order = torch.sort(draws, stable=True)                  # equal values stay in draw order
    rep_sorted = torch.zeros(m, dtype=torch.bool, device=device)
    rep_sorted[1:] = s[1:] == s[:-1]                           # a later draw of a value already drawn
    rep = torch.empty_like(rep_sorted)
    rep[order] = rep_sorted
    pos = torch.cumsum(~rep, 0) - 1                            # rank among the survivors, in draw order
    out = torch.zeros(k + 1, dtype=draws.dtype, device=device)
    out.scatter_(0, torch.where(~rep & (pos < k), pos, torch.full_like(pos, k)), draws)     # (slot k: the rest)
    return out[:k]


def sample_key_frame(depth, intr_mat, T_wc, sampling_size, ray_max_dist, generator=None, points=None):
    """IterableInferenceDataset._sample_key_frame (fusion_inference_dataset.py:373-420) for a depth map
    already on the device: ``sampling_size`` random pixels with their back-projected world points, validity
    and 3x3 neighbourhoods.  depth [H, W] metres; intr_mat [3, 3]; T_wc [4, 4] -> rays dict (batch 1).
    ``points``: the frame's ``key_frame_points`` when the caller keeps them (NeuralMap.optimize does, per key frame:
    the reference re-reads the depth image in DataLoader workers beside the optimiser, off its critical path)."""
    if points is not None:
        dev = points["pts"].device
        H, W = points["H"], points["W"]
        if generator is not None and generator.device.type == "cpu":
            idx = torch.randperm(H * W, generator=generator)[:sampling_size].to(dev)
        else:
            idx = random_subset(H * W, sampling_size, dev, generator)
        px, py = idx % W, idx // W
        nidx = (py[:, None] + points["oy"]).clamp(0, H - 1) * W + (px[:, None] + points["ox"]).clamp(0, W - 1)
        rgb = points["zero_rgb"].get(len(idx))        # (all zeros, read-only downstream: one tensor per batch size)
        if rgb is None:
            rgb = points["zero_rgb"][len(idx)] = torch.zeros(1, len(idx), 3, device=dev)
        return {"uv": torch.stack([px, py], -1).float().unsqueeze(0),
                "rgb": rgb,
                "gt_pts": points["pts"][idx].unsqueeze(0),
                "intr_mat": points["intr_mat"], "T_wc": points["T_wc"],
                "T_wc_host": points["T_wc_host"], "intr_host": points["intr_host"],
                "mask": points["mask"][idx].unsqueeze(0),
                "neighbor_pts": points["pts"][nidx].unsqueeze(0),
                "neighbor_masks": points["mask"][nidx].unsqueeze(0)}
    dev = depth.device
    intr_mat, T_wc = torch.as_tensor(intr_mat), torch.as_tensor(T_wc)
    depth = depth.to(torch.float64)
    mask = (depth > 0) & (depth < ray_max_dist)                       # common.py:110-113
    depth = depth * mask
    H, W = depth.shape
    K = intr_mat.to(dev, torch.float32)
    T = T_wc.to(dev, torch.float32).to(torch.float64)
    # geometry.py:163-168 forms the normalised pixel coordinates in float32 before the float64 product
    u = ((torch.arange(W, device=dev, dtype=torch.float32) - K[0, 2]) / K[0, 0]).to(torch.float64)
    v = ((torch.arange(H, device=dev, dtype=torch.float32) - K[1, 2]) / K[1, 1]).to(torch.float64)
    pts_c = torch.stack([u[None, :].expand(H, W), v[:, None].expand(H, W), torch.ones_like(depth)], -1)
    pts_c = pts_c * depth[..., None]                                  # geometry.py:150-171
    pts_w = pts_c.reshape(-1, 3) @ T[:3, :3].T + T[:3, 3]
    if generator is not None and generator.device.type == "cpu":
        idx = torch.randperm(H * W, generator=generator)[:sampling_size].to(dev)
    else:
        idx = random_subset(H * W, sampling_size, dev, generator)
    px, py = idx % W, idx // W
    uv = torch.stack([px, py], -1).float()
    r = torch.arange(-1, 2, device=dev)
    oy, ox = torch.meshgrid(r, r, indexing="ij")                      # np.meshgrid(range_, range_) order: x fastest
    nx = (px[:, None] + ox.reshape(-1)[None]).clamp(0, W - 1)
    ny = (py[:, None] + oy.reshape(-1)[None]).clamp(0, H - 1)
    nidx = ny * W + nx
    pts_map = pts_w
    return {
        "uv": uv.unsqueeze(0),
        "rgb": torch.zeros(1, len(idx), 3, device=dev),
        "gt_pts": pts_map[idx].float().unsqueeze(0),
        "intr_mat": intr_mat.to(dev).float().reshape(1, 3, 3),
        "T_wc": T_wc.to(dev).float().reshape(1, 4, 4),
        # host copies for the fused path (kernel arguments): free when the pose came from the host
        "T_wc_host": T_wc.detach().cpu().numpy().astype(np.float32).reshape(4, 4),
        "intr_host": intr_mat.detach().cpu().numpy().astype(np.float32).reshape(3, 3),
        "mask": mask.reshape(-1)[idx].float().unsqueeze(0),
        "neighbor_pts": pts_map[nidx].float().unsqueeze(0),
        "neighbor_masks": mask.reshape(-1)[nidx].float().unsqueeze(0),
    }


def ray_split_step(volume, rays, nerf, truncated_units, truncated_dist, ray_max_dist, sdf_delta=None,
                   generator=None, grad=None, return_pred=False):
    """calculate_loss + backward of one ray split with the fused kernels (csrc/rays.hip + the decode_pts
    forward / backward kernels): 6 launches instead of the ~180 of the torch formulation above, whose results
    it reproduces (same uniforms in the same order: fine strata first, then coarse).  ``d loss / d features`` is
    ACCUMULATED into ``grad`` ([M, 8], e.g. ``volume.features.grad``); returns (loss [1] device tensor, pts)."""
    lib = _lib.load()
    uv = rays["uv"][0].float().contiguous()
    dev = uv.device
    n = int(uv.shape[0])
    n_fine, n_coarse = int(truncated_units * 2), int(ray_max_dist * 5)
    S = n_fine + n_coarse
    u_f = _rand((1, n, n_fine), dev, generator).contiguous()
    u_c = _rand((1, n, n_coarse), dev, generator).contiguous()
    T = rays["T_wc_host"] if rays.get("T_wc_host") is not None else rays["T_wc"].detach().cpu().numpy()
    K = rays["intr_host"] if rays.get("intr_host") is not None else rays["intr_mat"].detach().cpu().numpy()
    T = (C.c_float * 16)(*np.asarray(T, dtype=np.float32).reshape(-1)[:16].tolist())
    K = (C.c_float * 9)(*np.asarray(K, dtype=np.float32).reshape(-1)[:9].tolist())
    gt = rays["gt_pts"][0].float().contiguous()
    rm = rays["mask"][0].float().contiguous()
    nb = rays["neighbor_pts"][0].float().contiguous()
    nbm = rays["neighbor_masks"][0].float().contiguous()
    pts = torch.empty((n, S, 3), dtype=torch.float32, device=dev)
    target = torch.empty((n, S), dtype=torch.float32, device=dev)
    weight = torch.empty((n, S), dtype=torch.float32, device=dev)
    _lib.check(lib.bnv_ray_samples(_lib.ptr(uv), _lib.ptr(gt), _lib.ptr(rm), _lib.ptr(nb), _lib.ptr(nbm),
                                   int(nb.shape[1]), T, K, _lib.ptr(u_f), _lib.ptr(u_c), n, n_fine, n_coarse,
                                   float(truncated_dist), _lib.ptr(pts), _lib.ptr(target), _lib.ptr(weight),
                                   _lib.stream_ptr()), "bnv_ray_samples")
    volume.count_optim_pts(pts)
    n_valid = (rm.sum() + 1e-4).reshape(1)
    pred = volume._decode_pts_forward(pts, nerf, sdf_delta, False, True).reshape(-1).contiguous()
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    g = torch.empty_like(pred)
    _lib.check(lib.bnv_ray_loss(_lib.ptr(pred), _lib.ptr(target), _lib.ptr(weight), _lib.ptr(n_valid), n * S,
                                _lib.ptr(loss), _lib.ptr(g), _lib.stream_ptr()), "bnv_ray_loss")
    if grad is not None:
        volume.decode_pts_backward(pts, nerf, g, grad)
    return (loss, pts, pred.view(n, S)) if return_pred else (loss, pts)


def ray_batch_step(volume, rays, nerf, truncated_units, truncated_dist, ray_max_dist, sdf_delta=None,
                   generator=None, grad=None, train_ray_splits=1000, return_pred=False):
    """ALL ray splits of one optimiser step at once (include/bnv_fusion.h: bnv_optim_step) -- what
    ``for lo in range(0, n_rays, train_ray_splits): ray_split_step(...)`` computes (run_e2e.py:127-153), in 5 launches
    instead of 6 per split: one sampling launch for every ray, count_optim of all splits recorded as per-row split masks,
    ONE forward + loss + backward kernel whose mask decisions see exactly the weights the split-by-split sequence would
    (weights[row] + 1 per split up to the query's own that touches the row), then the +1s applied.  The uniforms are drawn
    split by split in the reference's order when the generator is a CPU generator (bit-for-bit the reference's stream);
    with a device generator (or none) in two calls for the whole step.  ``d loss / d features`` of the SUM of the splits'
    losses is ACCUMULATED into ``grad``; returns (sum of the splits' losses [1], pts [n, S, 3][, pred [n, S]])."""
    lib = _lib.load()
    uv = rays["uv"][0].float().contiguous()
    dev = uv.device
    n = int(uv.shape[0])
    n_fine, n_coarse = int(truncated_units * 2), int(ray_max_dist * 5)
    S = n_fine + n_coarse
    per = int(train_ray_splits)
    n_splits = -(-n // per)
    if n_splits > 31:
        raise ValueError(f"{n_splits} ray splits in one step (at most 31: raise train_ray_splits)")
    if generator is not None and generator.device.type == "cpu":
        uf, uc = [], []
        for lo in range(0, n, per):                 # the reference's order: a split's fine strata, then its coarse ones
            k = min(per, n - lo)
            uf.append(torch.rand(1, k, n_fine, generator=generator))
            uc.append(torch.rand(1, k, n_coarse, generator=generator))
        u_f = torch.cat(uf, 1).to(dev).contiguous()
        u_c = torch.cat(uc, 1).to(dev).contiguous()
    else:
        u_f = torch.rand((1, n, n_fine), device=dev, generator=generator)
        u_c = torch.rand((1, n, n_coarse), device=dev, generator=generator)
    T = rays["T_wc_host"] if rays.get("T_wc_host") is not None else rays["T_wc"].detach().cpu().numpy()
    K = rays["intr_host"] if rays.get("intr_host") is not None else rays["intr_mat"].detach().cpu().numpy()
    T = (C.c_float * 16)(*np.asarray(T, dtype=np.float32).reshape(-1)[:16].tolist())
    K = (C.c_float * 9)(*np.asarray(K, dtype=np.float32).reshape(-1)[:9].tolist())
    gt = rays["gt_pts"][0].float().contiguous()
    rm = rays["mask"][0].float().contiguous()
    nb = rays["neighbor_pts"][0].float().contiguous()
    nbm = rays["neighbor_masks"][0].float().contiguous()
    pts = torch.empty((n, S, 3), dtype=torch.float32, device=dev)
    target = torch.empty((n, S), dtype=torch.float32, device=dev)
    weight = torch.empty((n, S), dtype=torch.float32, device=dev)
    _lib.check(lib.bnv_ray_samples(_lib.ptr(uv), _lib.ptr(gt), _lib.ptr(rm), _lib.ptr(nb), _lib.ptr(nbm),
                                   int(nb.shape[1]), T, K, _lib.ptr(u_f), _lib.ptr(u_c), n, n_fine, n_coarse,
                                   float(truncated_dist), _lib.ptr(pts), _lib.ptr(target), _lib.ptr(weight),
                                   _lib.stream_ptr()), "bnv_ray_samples")
    # per split: sum of its ray masks + 1e-4 (render_utils.py:553)
    if n % per == 0:
        n_valid = rm.view(n_splits, per).sum(1) + 1e-4
    else:
        n_valid = torch.stack([rm[lo: lo + per].sum() for lo in range(0, n, per)]) + 1e-4
    n_valid = n_valid.float().contiguous()
    split_samples = per * S
    volume.count_optim_splits(pts, split_samples)
    loss2 = torch.zeros(2, dtype=torch.float32, device=dev)
    pred = torch.empty((n, S), dtype=torch.float32, device=dev) if return_pred else None
    if grad is None:
        grad = torch.zeros_like(volume.features.detach())        # (the kernel needs somewhere to accumulate)
    if _lib.model_mode(nerf) == 2 or not hasattr(nerf, "sdf_bwd_pack"):
        # tiny-cuda-nn decoder: its own forward / backward kernels, all splits per launch
        p = volume.decode_pts_splits(pts, nerf, sdf_delta, split_samples)
        g = torch.empty_like(p)
        _lib.check(lib.bnv_ray_loss_splits(_lib.ptr(p), _lib.ptr(target), _lib.ptr(weight), _lib.ptr(n_valid), n * S,
                                           split_samples, _lib.ptr(loss2), _lib.ptr(g), _lib.stream_ptr()),
                   "bnv_ray_loss_splits")
        volume.decode_pts_backward_splits(pts, nerf, g, grad, split_samples)
        if pred is not None:
            pred.copy_(p.view(n, S))
    else:
        volume.optim_step(pts, nerf, sdf_delta, split_samples, target, weight, n_valid, loss2, grad, pred)
    volume.apply_split_counts()
    out = (loss2[:1], pts)
    return out + (pred,) if return_pred else out


def optimize_volume(volume, nerf, ray_batches, truncated_units, truncated_dist, ray_max_dist, sdf_delta=None,
                    train_ray_splits=1000, lr=0.001, generator=None, fused=True, batched=True):
    """NeuralMap.optimize (run_e2e.py:111-162): Adam on ``volume.features`` over an iterable of ray
    batches, ``train_ray_splits`` rays per backward, then the optimised features are written back into the
    hash volume.  Returns the list of per-iteration losses (device scalars).  ``fused``: the HIP ray kernels instead of
    the torch formulation; ``batched`` (with ``fused``): all splits of a step per launch (ray_batch_step) instead of split
    by split (ray_split_step) -- same decisions, same weights, gradients equal up to the order of float atomics."""
    volume.to_tensor()
    volume.features = torch.nn.Parameter(volume.features)
    # (one fused update kernel on the device instead of the foreach implementation's eight: the same formula)
    optimizer = torch.optim.Adam([volume.features], lr=lr, **({"fused": True} if volume.features.is_cuda else {}))
    history = []
    for rays in ray_batches:
        optimizer.zero_grad(set_to_none=False)
        if rays.get("T_wc_host") is not None:
            if np.isnan(rays["T_wc_host"]).any():
                continue
        elif torch.isnan(rays["T_wc"]).any():
            continue
        n_rays = rays["uv"].shape[1]
        total = None
        if fused and volume.features.grad is None:
            volume.features.grad = torch.zeros_like(volume.features)
        whole = ("T_wc", "intr_mat", "T_wc_host", "intr_host")
        if fused and batched and -(-n_rays // train_ray_splits) <= 31:
            # every split of the step in one set of launches (same mask decisions, same count_optim as split by split)
            loss, _ = ray_batch_step(volume, rays, nerf, truncated_units, truncated_dist, ray_max_dist,
                                     sdf_delta=sdf_delta, generator=generator, grad=volume.features.grad,
                                     train_ray_splits=train_ray_splits)
            optimizer.step()
            history.append(loss[0])
            continue
        for lo in range(0, n_rays, train_ray_splits):
            part = {k: (v[:, lo: lo + train_ray_splits] if k not in whole else v) for k, v in rays.items()}
            if fused:
                loss, _ = ray_split_step(volume, part, nerf, truncated_units, truncated_dist, ray_max_dist,
                                         sdf_delta=sdf_delta, generator=generator, grad=volume.features.grad)
                loss = loss[0]
            else:
                out = calculate_loss(volume, part, nerf, truncated_units, truncated_dist, ray_max_dist,
                                     sdf_delta=sdf_delta, generator=generator)
                loss = sum(v for k, v in out.items() if k[0] != "_")
                loss.backward()
            total = loss.detach() if total is None else total + loss.detach()
        optimizer.step()
        history.append(total)
    feats = volume.features.detach()
    volume.features = feats
    volume.insert(volume.active_coordinates, feats, volume.weights, volume.num_hits)
    return history
