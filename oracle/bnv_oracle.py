"""CPU oracle for the BNV-Fusion local-fusion + SDF-decode hot path.

TEST INFRASTRUCTURE ONLY.  This file restates, op for op in PyTorch-CPU fp32,
what the reference (likojack/bnv_fusion, mounted at /root/reference in the build
container) computes on the path SURVEY.md section 8 scopes.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package (``bnv_fusion_amd``) never does and has no CPU
fallback.

Parity pin: the reference has no tests of its own, so the
oracle is pinned against outputs of the reference itself, captured in the build
container by ``tests/golden/make_golden*.py`` (reference imported under the
shims of ``tests/golden/ref_shims.py``) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against every one of them
(voxel ids / counts bit-exact, floats to <= 1e-6; gradients and the optimiser's
ray loss from the reference's autograd; the TSDF side fusion from the
reference's CPU fallback).  NOT pinned, and said so where they are defined:
the tcnn (fp16 FullyFusedMLP) variant -- its reference arithmetic is CUDA-only
(layout verified statistically in SURVEY.md Appendix A only); the Sobel
normals of the depth front end (kornia absent); marching cubes (scikit-image
absent).

All file:line citations are relative to /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- #
# voxel index <-> flat id <-> world   (src/utils/voxel_utils.py)
# --------------------------------------------------------------------------- #


def get_world_range(dimensions, voxel_size):
    """voxel_utils.py:83-88 -- pads one voxel each side; float64 numpy."""
    dimensions = np.asarray(dimensions, dtype=np.float64)
    min_ = -dimensions / 2 - voxel_size
    max_ = dimensions / 2 + voxel_size
    n_xyz = np.ceil((max_ - min_) / voxel_size).astype(int).tolist()
    max_ = min_ + voxel_size * np.asarray(n_xyz)
    return min_, max_, n_xyz


def flatten(voxels, n_xyz):
    """voxel_utils.py:62-65."""
    return voxels[..., 0] * n_xyz[1] * n_xyz[2] + voxels[..., 1] * n_xyz[2] + voxels[..., 2]


def unflatten(flat_id, n_xyz):
    """voxel_utils.py:68-80 (torch branch)."""
    x = torch.div(flat_id, (n_xyz[1] * n_xyz[2]), rounding_mode="floor")
    rest = flat_id % (n_xyz[1] * n_xyz[2])
    y = torch.div(rest, n_xyz[2], rounding_mode="floor")
    z = flat_id - x * n_xyz[1] * n_xyz[2] - y * n_xyz[2]
    return torch.stack([x, y, z], dim=-1)


_CORNER_IS_CEIL = (  # (x, y, z) uses ceil?  -- the order of modules.py:590-655 / fusion/utils.py:98-167
    (0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1))


def get_neighbors(points, as_int=False):
    """points [b, n, s, 3] -> [b, 8, n, s, 3].

    ``as_int=True``  : ReplicateNeRFModel.get_neighbors, modules.py:586-655 (``.int()``)
    ``as_int=False`` : fusion/utils.py:98-167 (float corners, used by SparseVolume.decode_pts)
    """
    fl, ce = torch.floor(points), torch.ceil(points)
    out = []
    for cx, cy, cz in _CORNER_IS_CEIL:
        out.append(torch.stack([(ce if cx else fl)[..., 0], (ce if cy else fl)[..., 1],
                                (ce if cz else fl)[..., 2]], dim=-1))
    out = torch.stack(out, dim=1)
    return out.int() if as_int else out


# --------------------------------------------------------------------------- #
# networks
# --------------------------------------------------------------------------- #


def load_weights(npz_path):
    """Converted checkpoint (tests/golden/convert_checkpoints.py) -> dict of torch tensors."""
    with np.load(npz_path) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def pointnet_encoder(sd, x):
    """PointNetEncoder.forward(x, global_feat=False), pointnet_utils.py:246-266.
    x [B, 6, P] -> [B, 8, P].  BatchNorm1d in eval mode (run_e2e.py:234)."""
    p = "pointnet_backbone."
    for i in (1, 2, 3, 4):
        x = F.conv1d(x, sd[f"{p}conv{i}.weight"], sd[f"{p}conv{i}.bias"])
        x = F.batch_norm(x, sd[f"{p}bn{i}.running_mean"], sd[f"{p}bn{i}.running_var"],
                         sd[f"{p}bn{i}.weight"], sd[f"{p}bn{i}.bias"], False, 0.1, 1e-5)
        if i < 4:
            x = F.relu(x)
    return x


def xyz_encoding(t):
    """positional_encoding(num_encoding_functions=1, include_input=True), modules.py:81-123.
    [..., 3] -> [..., 9] = [t, sin(t * 1.0), cos(t * 1.0)]."""
    freq = 2.0 ** torch.linspace(0.0, 0.0, 1, dtype=t.dtype)
    enc = [t]
    for f in freq:
        for fn in (torch.sin, torch.cos):
            enc.append(fn(t * f))
    return torch.cat(enc, dim=-1)


def geo_forward(sd, x, num_layers=4):
    """ReplicateNeRFModel.geo_forward, modules.py:657-662.  [..., 17] -> [..., 1]."""
    for i in range(num_layers):
        x = F.relu(F.linear(x, sd[f"nerf.geo_layer{i}.weight"], sd[f"nerf.geo_layer{i}.bias"]))
    return F.linear(x, sd["nerf.fc_alpha.weight"], sd["nerf.fc_alpha.bias"])


def forward_with_mask(sd, input_feats, mask):
    """modules.py:774-783 -- MLP on the masked rows only, zeros elsewhere."""
    shapes = list(input_feats.shape)
    mask = mask.reshape(-1)
    flat = input_feats.reshape(-1, shapes[-1])
    alpha = geo_forward(sd, flat[mask])
    out = torch.zeros_like(flat[:, :1], dtype=alpha.dtype)
    out[mask] = alpha
    return out.reshape(shapes[:-1] + [1])


# --------------------------------------------------------------------------- #
# encode  (LitFusionPointNet.encode_pointcloud, local_point_fusion.py:81-165)
# --------------------------------------------------------------------------- #


def get_relative_xyz(xyz, bound_min, voxel_size):
    """local_point_fusion.py:153-165."""
    xyz_zeroed = xyz - bound_min
    xyz_normalized = xyz_zeroed / voxel_size
    grid_id = get_neighbors(xyz_normalized.unsqueeze(1), as_int=True).squeeze(2)  # [B, 8, N, 3] i32
    relative = (xyz_normalized.unsqueeze(1) - grid_id) * voxel_size
    return relative, grid_id


def encode_pointcloud(sd, input_pts, n_xyz, bound_min, bound_max, voxel_size,
                      min_pts_in_grid=8, return_dense=False, encoder=None):
    """local_point_fusion.py:81-151.  input_pts [1, N, 6] f32.
    sparse: (feats [U',8], pcounts [U',1] i64, flat_ids [U'] i64, grid_ids [U',3] i64, n_avg_pts)
    dense : (feat_grids [1,8,X,Y,Z], mask [1,1,X,Y,Z], unique_flat_ids [U], flat_ids [1,P])."""
    res = [int(v) for v in n_xyz]
    in_xyz = input_pts[:, :, :3] * 1.
    in_normal = input_pts[:, :, 3:]
    bound_mask = (in_xyz[0, :, 0] < bound_max[0] - voxel_size) \
        * (in_xyz[0, :, 1] < bound_max[1] - voxel_size) \
        * (in_xyz[0, :, 2] < bound_max[2] - voxel_size) \
        * (in_xyz[0, :, 0] > bound_min[0] + voxel_size) \
        * (in_xyz[0, :, 1] > bound_min[1] + voxel_size) \
        * (in_xyz[0, :, 2] > bound_min[2] + voxel_size)
    if torch.sum(bound_mask) == 0:
        return None, None, None, None, None
    in_xyz = in_xyz[:, bound_mask, :]
    in_normal = in_normal[:, bound_mask, :]
    relative_xyz, grid_id = get_relative_xyz(in_xyz, bound_min, voxel_size)
    grid_id = grid_id.reshape(1, -1, 3)
    pointnet_input = torch.cat([relative_xyz, in_normal.unsqueeze(1).repeat(1, 8, 1, 1)], dim=-1)
    pointnet_input = pointnet_input.reshape(1, -1, 6)
    # forward(normalize=True), local_point_fusion.py:51-65
    pointnet_input[:, :, :3] = pointnet_input[:, :, :3] / voxel_size
    assert torch.min(pointnet_input[:, :, :3]) >= -1 and torch.max(pointnet_input[:, :, :3]) <= 1
    point_feats = (encoder or (lambda x: pointnet_encoder(sd, x)))(pointnet_input.permute(0, 2, 1))  # [1, 8, P]
    flat_ids = flatten(grid_id, n_xyz).long()
    unique_flat_ids, pinds, pcounts = torch.unique(flat_ids[0], return_inverse=True, return_counts=True)
    unique_grid_ids = unflatten(unique_flat_ids, n_xyz).long()
    assert torch.max(unique_grid_ids[:, 0]) < res[0] and torch.max(unique_grid_ids[:, 1]) < res[1]
    assert torch.max(unique_grid_ids[:, 2]) < res[2] and torch.min(unique_grid_ids) >= 0
    # torch_scatter.scatter_mean (local_point_fusion.py:125): sum / max(count, 1)
    U = unique_flat_ids.shape[0]
    idx = pinds[None, None, :].expand_as(point_feats)
    total = torch.zeros((1, point_feats.shape[1], U), dtype=point_feats.dtype).scatter_add_(2, idx, point_feats)
    count = torch.zeros_like(total).scatter_add_(2, idx, torch.ones_like(point_feats))
    point_feats_mean = total / count.clamp(min=1)
    point_feats_mean[:, :, pcounts < min_pts_in_grid] = 0
    if return_dense:
        feat_grids = torch.zeros((1, point_feats.shape[1], *res), dtype=point_feats.dtype)
        mask = torch.zeros((1, 1, *res), dtype=point_feats.dtype)
        g = unique_grid_ids
        mask[0, 0, g[:, 0], g[:, 1], g[:, 2]] = pcounts.type(point_feats.type())
        feat_grids[0, :, g[:, 0], g[:, 1], g[:, 2]] = point_feats_mean
        return feat_grids, mask, unique_flat_ids, flat_ids
    n_avg_pts = torch.mean(pcounts.type(point_feats.type()))
    valid = pcounts >= min_pts_in_grid
    point_feats_mean = point_feats_mean[:, :, valid]
    pcounts = pcounts[valid]
    unique_flat_ids = unique_flat_ids[valid]
    unique_grid_ids = unflatten(unique_flat_ids, n_xyz).long()
    return point_feats_mean[0].permute(1, 0), pcounts.unsqueeze(-1), unique_flat_ids, unique_grid_ids, n_avg_pts


# --------------------------------------------------------------------------- #
# sparse volume  (SparseVolume, sparse_volume.py:484-695)
# --------------------------------------------------------------------------- #


class OracleSparseVolume:
    """Semantics of sparse_volume.py:484-695 with the Open3D hash map replaced by a
    Python dict (key tuple -> buffer row, insertion order = buffer order)."""

    def __init__(self, n_feats, voxel_size, dimensions, min_pts_in_grid):
        min_coords, max_coords, n_xyz = get_world_range(dimensions, voxel_size)
        self.dimensions = dimensions
        self.voxel_size = voxel_size
        self.min_coords = torch.from_numpy(min_coords).float()   # sparse_volume.py:495
        self.max_coords = torch.from_numpy(max_coords).float()
        self.n_xyz = torch.from_numpy(np.asarray(n_xyz)).long()  # sparse_volume.py:497
        self.n_feats = n_feats
        self.min_pts_in_grid = min_pts_in_grid
        self._map = {}
        self._keys, self._feats, self._w, self._hits = [], [], [], []
        self.features = self.weights = self.num_hits = self.active_coordinates = None
        self._tensor_map = None
        self.n_pts_list = []

    def track_n_pts(self, n_pts):
        """sparse_volume.py:508-513."""
        self.n_pts_list.append(float(n_pts))

    def _rows(self, keys, table):
        kl = keys.reshape(-1, 3).long().tolist()
        return [table.get(tuple(k), -1) for k in kl]

    def query(self, keys):
        """sparse_volume.py:661-695: zeros for absent keys."""
        shapes = list(keys.shape)
        n = int(np.prod(shapes[:-1]))
        if n == 0:
            return None, None, None
        rows = self._rows(keys, self._map)
        f = torch.zeros((n, self.n_feats))
        w = torch.zeros((n, 1))
        h = torch.zeros((n, 1))
        for i, r in enumerate(rows):
            if r >= 0:
                f[i], w[i], h[i] = self._feats[r], self._w[r], self._hits[r]
        return (f.reshape(shapes[:-1] + [self.n_feats]), w.reshape(shapes[:-1] + [1]),
                h.reshape(shapes[:-1] + [1]))

    def insert(self, keys, new_feats, new_weights, new_num_hits):
        """sparse_volume.py:561-585: upsert (new keys appended, existing keys overwritten)."""
        if len(keys) == 0:
            return None
        kl = keys.reshape(-1, 3).long().tolist()
        for i, k in enumerate(kl):
            k = tuple(k)
            r = self._map.get(k)
            if r is None:
                self._map[k] = len(self._keys)
                self._keys.append(k)
                self._feats.append(new_feats[i].detach().clone())
                self._w.append(new_weights[i].detach().clone())
                self._hits.append(new_num_hits[i].detach().clone())
            else:
                self._feats[r] = new_feats[i].detach().clone()
                self._w[r] = new_weights[i].detach().clone()
                self._hits[r] = new_num_hits[i].detach().clone()

    def to_tensor(self):
        """sparse_volume.py:525-559: snapshot of the active entries + key -> row index."""
        self.active_coordinates = torch.tensor(self._keys, dtype=torch.int64).reshape(-1, 3)
        self.features = torch.stack(self._feats) if self._feats else torch.zeros((0, self.n_feats))
        self.weights = torch.stack(self._w).reshape(-1, 1) if self._w else torch.zeros((0, 1))
        self.num_hits = torch.stack(self._hits).reshape(-1, 1) if self._hits else torch.zeros((0, 1))
        self._tensor_map = {k: i for i, k in enumerate(self._keys)}
        return self.active_coordinates, self.features, self.weights, self.num_hits

    def _query_tensor(self, keys):
        """sparse_volume.py:625-659: lookup in the to_tensor() snapshot."""
        shapes = list(keys.shape)
        n = int(np.prod(shapes[:-1]))
        rows = torch.tensor(self._rows(keys, self._tensor_map), dtype=torch.int64)
        found = rows >= 0
        f = torch.zeros((n, self.n_feats))
        w = torch.zeros((n, 1))
        h = torch.zeros((n, 1))
        f[found] = self.features[rows[found]]
        w[found] = self.weights[rows[found]]
        h[found] = self.num_hits[rows[found]]
        return (f.reshape(shapes[:-1] + [self.n_feats]), w.reshape(shapes[:-1] + [1]),
                h.reshape(shapes[:-1] + [1]))

    def count_optim(self, keys):
        """sparse_volume.py:602-622: weights[row] += 1 for found keys (index_put, no accumulate:
        duplicated rows are incremented once)."""
        rows = torch.tensor(self._rows(keys, self._tensor_map), dtype=torch.int64)
        rows = rows[rows >= 0]
        self.weights[rows] += 1

    def decode_pts(self, coords, sd, sdf_delta=None, is_coords=False, query_tensor=True, geo=None):
        return decode_pts(self, coords, sd, sdf_delta, is_coords, query_tensor, geo)


def integrate(volume, fine_coords, fine_feats, fine_weights):
    """LitFusionPointNet._integrate + _update, local_point_fusion.py:647-673."""
    fine_weights = torch.clip(fine_weights / 32, max=1)
    old_f, old_w, hits = volume.query(fine_coords)
    new_f = new_w = None
    if len(fine_coords) > 0:
        new_w = old_w + fine_weights
        new_f = (old_f * old_w + fine_feats * fine_weights) / new_w
    volume.insert(fine_coords, new_f, new_w, hits)


# --------------------------------------------------------------------------- #
# decode
# --------------------------------------------------------------------------- #


def decode_pts(volume, coords, sd, sdf_delta=None, is_coords=False, query_tensor=True, geo=None):
    """SparseVolume.decode_pts, sparse_volume.py:768-833.  coords [1, B, S, 3] -> [1, B, S, 1].
    ``geo``: alternative nerf.geo_forward (tcnn variant); its fp16 output keeps ``alpha * voxel_size`` in
    fp16 as torch does for a half tensor times a python float."""
    if not is_coords:
        coords = (coords - volume.min_coords) / volume.voxel_size
    neighbor_coords = get_neighbors(coords)                      # float corners [1, 8, B, S, 3]
    local_coords = coords.unsqueeze(1) - neighbor_coords
    assert torch.min(local_coords) >= -1 and torch.max(local_coords) <= 1
    weights_unmasked = torch.prod(1 - torch.abs(local_coords), dim=-1, keepdim=True)
    if query_tensor:
        feats, weights, _ = volume._query_tensor(neighbor_coords)
    else:
        feats, weights, _ = volume.query(neighbor_coords)
    mask = torch.min(weights, dim=1)[0] >= volume.min_pts_in_grid
    nerf_in = torch.cat([xyz_encoding(local_coords), feats], dim=-1)
    if geo is None:
        alpha = geo_forward(sd, nerf_in)
        alpha = alpha * volume.voxel_size
    else:
        alpha = (geo(nerf_in).half() * volume.voxel_size).float()
    normalizer = torch.sum(weights_unmasked, dim=1, keepdim=True)
    weights_unmasked = weights_unmasked / normalizer
    alpha = torch.sum(alpha * weights_unmasked, dim=1)
    alpha = torch.where(mask, alpha, torch.zeros_like(alpha) + volume.voxel_size)
    if sdf_delta is not None:
        g = neighbor_coords / (volume.n_xyz - 1)
        g = (g * 2 - 1)[..., [2, 1, 0]]
        d = F.grid_sample(sdf_delta, g, mode="nearest", padding_mode="zeros", align_corners=True)
        d = d.permute(0, 2, 3, 4, 1)
        alpha = alpha + torch.sum(d * weights_unmasked, dim=1)
    return alpha


def lattice_coords(origins, step_size=0.5):
    """The 3x3x3 per-voxel lattice of SparseVolume.meshlize, sparse_volume.py:718-731.
    origins [B, 3] integer voxel coords -> [1, B, 27, 3] f32 (float64 numpy, then .float())."""
    origin = np.asarray(origins, dtype=np.int64)
    range_ = np.arange(0, 1 + step_size, step_size) - 0.5
    vc = np.stack(np.meshgrid(range_, range_, range_, indexing="ij"), axis=-1)
    vc = np.tile(vc, (len(origin), 1, 1, 1, 1))
    vc += origin[:, None, None, None, :]
    return torch.from_numpy(vc).float().reshape(1, len(origin), -1, 3)


def decode_feature_grid_w_pts(sd, voxel_coords, feat_grid, pts_weight, voxel_size, min_pts_in_grid=8,
                              global_coords=False, interpolate_decode=True):
    """LitFusionPointNet.decode_feature_grid_w_pts, local_point_fusion.py:265-367 (+ decode_implicit :372-379,
    LocalNeRFModel.forward(test=True) modules.py:941-960).  voxel_coords [1, Q, 3] -> sdf [1, Q], neighbor_feats.
    Defaults = the yaml configuration (global_coords=False, interpolate_decode=True, :281-329); the other two
    branches: nearest voxel (:288-292, 331-343) and global coordinates (:345-367, the signature default)."""
    h, w, d = feat_grid.shape[-3:]
    res = torch.tensor([h, w, d])
    if global_coords:
        g = voxel_coords / (res - 1)
        g = (g * 2 - 1)[..., [2, 1, 0]].unsqueeze(0).unsqueeze(0)           # [1, 1, 1, Q, 3]
        nf = F.grid_sample(feat_grid, g, mode="bilinear", padding_mode="zeros", align_corners=True)
        nf = nf.squeeze(2).squeeze(2).permute(0, 2, 1)                     # [1, Q, F]
        pw = F.grid_sample(pts_weight, g, mode="nearest", padding_mode="zeros", align_corners=True)
        pw = pw.squeeze(2).squeeze(2).permute(0, 2, 1)[..., 0]             # [1, Q]
        pts = voxel_coords / (res - 1)                                     # decode_implicit(normalize=False)
        sdf = geo_forward(sd, torch.cat([xyz_encoding(pts[..., :3]), nf], dim=-1))[..., 0]
        sdf = torch.where(pw >= min_pts_in_grid, sdf, torch.ones_like(sdf) * voxel_size)
        return sdf, nf
    if not interpolate_decode:
        nc = torch.round(voxel_coords)
        g = nc / (res - 1)
        g = (g * 2 - 1)[..., [2, 1, 0]].unsqueeze(0).unsqueeze(0)
        nf = F.grid_sample(feat_grid, g, mode="nearest", padding_mode="zeros", align_corners=True)
        pw = F.grid_sample(pts_weight, g, mode="nearest", padding_mode="zeros", align_corners=True)
        pw = pw * (pw >= min_pts_in_grid)
        nf = nf.squeeze(2).squeeze(2).permute(0, 2, 1)                     # [1, Q, F]
        pw = pw.squeeze(2).squeeze(2).permute(0, 2, 1)[..., 0]             # [1, Q]
        rel_xyz = (voxel_coords - nc) * voxel_size
        pts = rel_xyz / voxel_size                                         # decode_implicit(normalize=True)
        geo_in = torch.cat([xyz_encoding(pts[..., :3]), nf], dim=-1)
        sdf = (forward_with_mask(sd, geo_in, pw >= min_pts_in_grid) * voxel_size)[..., 0]
        sdf = torch.where(pw > 0, sdf, torch.ones_like(sdf) * voxel_size)
        return sdf, nf
    neighbor_coords = get_neighbors(voxel_coords.unsqueeze(1), as_int=True).squeeze(2)  # [1, 8, Q, 3] i32
    g = neighbor_coords / (res - 1)
    g = (g * 2 - 1)[..., [2, 1, 0]].unsqueeze(0)
    nf = F.grid_sample(feat_grid, g, mode="nearest", padding_mode="zeros", align_corners=True)
    pw = F.grid_sample(pts_weight, g, mode="nearest", padding_mode="zeros", align_corners=True)
    pw = pw * (pw >= min_pts_in_grid)
    nf = nf.squeeze(2).permute(0, 2, 3, 1)          # [1, 8, Q, F]
    pw = pw.squeeze(2).permute(0, 2, 3, 1)[..., 0]  # [1, 8, Q]
    rel = voxel_coords.unsqueeze(1) - neighbor_coords
    bw = torch.prod(1 - torch.abs(rel), dim=-1)
    bw = bw / torch.sum(bw, dim=1, keepdim=True)
    rel_xyz = rel * voxel_size
    pts = rel_xyz / voxel_size                       # decode_implicit(normalize=True)
    geo_in = torch.cat([xyz_encoding(pts[..., :3]), nf], dim=-1)
    sdf = forward_with_mask(sd, geo_in, pw >= min_pts_in_grid) * voxel_size
    sdf = torch.sum(sdf[..., 0] * bw, dim=1)
    valid = torch.sum(pw, dim=1) > 0
    sdf = torch.where(valid, sdf, torch.ones_like(sdf) * voxel_size)
    return sdf, nf


# --------------------------------------------------------------------------- #
# tcnn (FullyFusedMLP) variant -- PARITY UNPINNED (see module docstring)
# --------------------------------------------------------------------------- #


def tcnn_mlp(params, x, n_in_padded, n_out, width=64, n_hidden=3, half=True, ste=False, acc16=False):
    """tiny-cuda-nn FullyFusedMLP restated: identity encoding pads the input to a multiple of 16
    with 1.0; ``n_hidden`` hidden layers of ``width`` with ReLU, no bias, output padded to 16;
    weights row-major [out, in] in one flat vector (call sites
    pointnet_utils.py:274-279, modules.py:171-176).  ``half`` rounds weights, inputs and every
    layer output to fp16 as the CUDA kernel stores them (fp32 accumulate).
    ``acc16``: a SENSITIVITY variant, not a parity claim -- the accumulator is rounded to fp16 after every 16-deep
    step of the contraction, what a kernel that keeps its accumulator fragments in half precision does [from memory of
    tiny-cuda-nn's fully_fused_mlp.cu: wmma accumulator fragments of __half; the CUDA source is not in the image and
    cannot be run here].  The HIP kernels accumulate in fp32 like the default; the gap between the two variants bounds
    what that unverifiable detail could change (tests/test_oracle_golden.py)."""
    q = (lambda t: t.half().float()) if half else (lambda t: t)
    if half and ste:
        # for gradient checks: the fp16 rounding counts as identity in the backward pass (what autograd does
        # for a dtype cast) WITHOUT rounding the gradient itself to fp16
        q = lambda t: t + (t.half().float() - t).detach()
    n = x.shape[0]
    pad = torch.ones((n, n_in_padded - x.shape[1]), dtype=x.dtype)
    h = q(torch.cat([x, pad], dim=1))
    dims = [n_in_padded] + [width] * n_hidden + [16]
    off = 0
    for i in range(len(dims) - 1):
        w = q(params[off: off + dims[i + 1] * dims[i]].reshape(dims[i + 1], dims[i]))
        off += dims[i + 1] * dims[i]
        if acc16:
            acc = torch.zeros((n, dims[i + 1]), dtype=h.dtype)
            for k0 in range(0, dims[i], 16):
                acc = (acc + h[:, k0: k0 + 16] @ w[:, k0: k0 + 16].t()).half().float()
            h = acc
        else:
            h = h @ w.t()
        if i < len(dims) - 2:
            h = F.relu(h)
        h = q(h)
    return h[:, :n_out]


def tcnn_point_encoder(params, acc16=False):
    """tcnnPointNetEncoder.forward(x, global_feat=False), pointnet_utils.py:283-294: x [1, 6, P] -> [1, 8, P]."""
    return lambda x: tcnn_mlp(params, x[0].t(), 16, 8, acc16=acc16).t()[None]


def tcnn_geo_forward(params, ste=False, acc16=False):
    """tcnnNeRFModel.geo_forward, modules.py:249-253: [..., 17] -> [..., 1]."""
    def f(x):
        shp = list(x.shape)
        return tcnn_mlp(params, x.reshape(-1, shp[-1]), 32, 1, ste=ste, acc16=acc16).reshape(shp[:-1] + [1])
    return f


# --------------------------------------------------------------------------- #
# synthetic benchmark frames -- dataset path restated
# --------------------------------------------------------------------------- #


def depth_to_input_pts(depth, intr, T_wc, max_depth=10.0):
    """FusionInferenceAbstractDataset.__getitem__, fusion_inference_dataset.py:40-90 -> [N, 6] float64.

    * points: geometry.depth2xyz (geometry.py:150-171) -- the pixel rays are FLOAT32 there
      (``np.arange(width, dtype=np.float32) - cx) / fx`` under the reference's numpy-1.x value-based
      casting), then float64 x depth, then T_wc @ homogeneous (fusion_inference_dataset.py:68-69);
    * normals: kornia-0.6.2 depth_to_normals (:52-55), which is absent here and restated as in
      geometry.py:515-527 (PARITY UNPINNED against kornia itself): float64 xyz map -> 3x3 Sobel / 8 with
      replicate padding -> cross(d/du, d/dv) -> L2 normalise (eps 1e-12) -> rotated by T_wc[:3,:3] (:66);
    * mask: 0 < depth < max_depth (common.py:107-110), rows kept in row-major pixel order (:74).
    Matrix products are written out as mul/add chains so that the HIP kernel can follow the same
    float64 operation order."""
    depth = np.asarray(depth, dtype=np.float64)
    mask = (depth > 0) & (depth < max_depth)
    depth = depth * mask                                  # load_depth zeroes masked pixels (common.py:109-110)
    H, W = depth.shape
    fx, fy, cx, cy = intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2]
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    xyz = np.stack([(u - cx) / fx * depth, (v - cy) / fy * depth, depth], axis=0)  # [3, H, W]
    p = np.pad(xyz, ((0, 0), (1, 1), (1, 1)), mode="edge")
    gx = (((((p[:, :-2, 2:] + 2 * p[:, 1:-1, 2:]) + p[:, 2:, 2:]) - p[:, :-2, :-2]) - 2 * p[:, 1:-1, :-2])
          - p[:, 2:, :-2]) / 8.0
    gy = (((((p[:, 2:, :-2] + 2 * p[:, 2:, 1:-1]) + p[:, 2:, 2:]) - p[:, :-2, :-2]) - 2 * p[:, :-2, 1:-1])
          - p[:, :-2, 2:]) / 8.0
    n = np.stack([gx[1] * gy[2] - gx[2] * gy[1], gx[2] * gy[0] - gx[0] * gy[2], gx[0] * gy[1] - gx[1] * gy[0]])
    norm = np.sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2])
    n = n / np.maximum(norm, 1e-12)
    ur = ((np.arange(W, dtype=np.float32) - np.float32(cx)) / np.float32(fx)).astype(np.float64)
    vr = ((np.arange(H, dtype=np.float32) - np.float32(cy)) / np.float32(fy)).astype(np.float64)
    pc = np.stack([ur[None, :] * depth, vr[:, None] * depth, depth], axis=0)
    T = np.asarray(T_wc, dtype=np.float64)
    pw = [((T[i, 0] * pc[0] + T[i, 1] * pc[1]) + T[i, 2] * pc[2]) + T[i, 3] for i in range(3)]
    nw = [(T[i, 0] * n[0] + T[i, 1] * n[1]) + T[i, 2] * n[2] for i in range(3)]
    out = np.stack(pw + nw, axis=-1).reshape(-1, 6)
    return out[mask.reshape(-1)]


def tsdf_integrate(tsdf, weight, vol_origin, voxel_size, depth_im, cam_intr, cam_pose, obs_weight=1.0,
                   cpu_path=False):
    """TSDFVolume.integrate (third_parties/fusion.py).  tsdf / weight [X, Y, Z] float32 are updated in place;
    colour is not restated.

    Default: as the inline CUDA kernel computes it (:68-126) -- float32 throughout, un-fused multiply-adds,
    transposed rotation, ``roundf`` (half away from zero) for the pixel; this is what the reference runs on a GPU
    box and what csrc/tsdf.hip follows.  ``cpu_path=True``: as the reference's CPU fallback computes it
    (:283-305) -- camera points through ``np.linalg.inv(cam_pose)`` in float64, ``np.round`` (half to even).
    PIN: tests/golden/tsdf_40.npz was captured from that CPU fallback (make_golden_tsdf.py; numba shimmed as plain
    Python); ``cpu_path=True`` reproduces it, and the default differs from it only in the ~1 % of voxels whose
    projection falls within rounding distance of a pixel boundary."""
    f = np.float32
    X, Y, Z = tsdf.shape
    vx, vy, vz = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    org = np.asarray(vol_origin, dtype=f)
    vs = f(voxel_size)
    trunc = f(5 * float(voxel_size))
    K = np.asarray(cam_intr, dtype=np.float64)[:3, :3].astype(f)
    P = np.asarray(cam_pose, dtype=np.float64).astype(f)
    pt = [org[i] + v.astype(f) * vs for i, v in enumerate((vx, vy, vz))]
    if cpu_path:
        # vox2world (float32) -> rigid_transform with the inverted pose (float64) -> cam2pix (fusion.py:171-195)
        Tinv = np.linalg.inv(np.asarray(cam_pose))
        xyz = np.stack([p.astype(np.float32) for p in pt], -1).reshape(-1, 3)
        cam_all = (Tinv @ np.hstack([xyz, np.ones((len(xyz), 1), dtype=np.float32)]).T).T[:, :3]
        cam = [cam_all[:, i].reshape(X, Y, Z) for i in range(3)]
        with np.errstate(divide="ignore", invalid="ignore"):
            ux = cam[0] * K[0, 0] / cam[2] + K[0, 2]
            uy = cam[1] * K[1, 1] / cam[2] + K[1, 2]
        rnd = lambda a: np.where(np.isfinite(a), np.round(a), -1).astype(np.int64)
        px, py = rnd(ux), rnd(uy)
        H, W = depth_im.shape
        ok = (px >= 0) & (px < W) & (py >= 0) & (py < H) & (cam[2] > 0)
        d = np.zeros(cam[2].shape)
        d[ok] = np.asarray(depth_im)[py[ok], px[ok]]
        diff = d - cam[2]
        ok = (d > 0) & (diff >= -float(trunc))
        dist = np.minimum(1, diff / float(trunc))
        w_old = weight[ok]
        w_new = (w_old + f(obs_weight)).astype(f)
        tsdf[ok] = ((w_old * tsdf[ok] + f(obs_weight) * dist[ok].astype(f)) / w_new).astype(f)
        weight[ok] = w_new
        return tsdf, weight
    t = [pt[i] - P[i, 3] for i in range(3)]
    cam = [(P[0, i] * t[0] + P[1, i] * t[1]) + P[2, i] * t[2] for i in range(3)]
    with np.errstate(divide="ignore", invalid="ignore"):
        ux = K[0, 0] * (cam[0] / cam[2]) + K[0, 2]
        uy = K[1, 1] * (cam[1] / cam[2]) + K[1, 2]
    rnd = lambda a: np.where(np.isfinite(a), np.sign(a) * np.floor(np.abs(a) + f(0.5)), -1).astype(np.int64)  # roundf
    px, py = rnd(ux), rnd(uy)
    H, W = depth_im.shape
    ok = (px >= 0) & (px < W) & (py >= 0) & (py < H) & ~(cam[2] < 0)
    d = np.zeros_like(cam[2])
    d[ok] = np.asarray(depth_im, dtype=f)[py[ok], px[ok]]
    ok &= d != 0
    diff = d - cam[2]
    ok &= ~(diff < -trunc)
    dist = np.minimum(f(1.0), diff / trunc)
    w_old = weight[ok]
    w_new = w_old + f(obs_weight)
    weight[ok] = w_new
    tsdf[ok] = (tsdf[ok] * w_old + f(obs_weight) * dist[ok]) / w_new
    return tsdf, weight


# --------------------------------------------------------------------------- #
# global optimiser edge: ray sampling + SDF loss, src/utils/render_utils.py
# Pinned by tests/golden/optimize_64.npz / decode_grad_64.npz (make_golden_grad.py).  Gradients come
# from torch autograd through decode_pts above (volume.features with requires_grad).
# --------------------------------------------------------------------------- #


def camera_rays(uv, T_wc, intr):
    """get_camera_params + lift for a 4x4 pose (render_utils.py:411-458): unit ray directions [b, n, 3]
    through pixel coordinates uv [b, n, 2] and the camera centre [b, 3]."""
    fx, fy = intr[:, 0, 0, None], intr[:, 1, 1, None]
    cx, cy, sk = intr[:, 0, 2, None], intr[:, 1, 2, None], intr[:, 0, 1, None]
    x, y = uv[..., 0], uv[..., 1]
    z = x * 0.0 + 1.0
    xl = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    yl = (y - cy) / fy * z
    cam = torch.stack([xl, yl, z, torch.ones_like(z)], dim=-1)             # [b, n, 4]
    world = torch.bmm(T_wc, cam.permute(0, 2, 1)).permute(0, 2, 1)[..., :3]
    centre = T_wc[:, :3, 3]
    return F.normalize(world - centre[:, None, :], dim=2), centre


def stratified_samples(n_samples, lengths, rand):
    """stratified_sampling (render_utils.py:77-94): one uniform draw inside each of n_samples strata of
    [0, length]; ``lengths`` [b, n, 1]; ``rand(b, n, s)`` supplies the uniforms -> [b, n, s, 1]."""
    b, n = lengths.shape[:2]
    edges = torch.linspace(0, 1, steps=n_samples).unsqueeze(0).repeat(b, n, 1) * lengths
    mids = 0.5 * (edges[..., 1:] + edges[..., :-1])
    upper = torch.cat([mids, edges[..., -1:]], dim=-1)
    lower = torch.cat([edges[..., :1], mids], dim=-1)
    return (lower + (upper - lower) * rand(b, n, n_samples)).unsqueeze(-1)


def hierarchical_samples(n_fine, n_coarse, depths, surface, dirs, centre, offset, rand):
    """hierarchical_sampling (render_utils.py:191-233): n_fine samples in [depth - offset, depth + offset]
    around the observed surface + n_coarse along the whole ray up to the surface, sorted by distance."""
    back = torch.where(depths - offset < 0, depths, torch.zeros_like(depths) + offset)
    start = surface - back.unsqueeze(-1) * dirs
    start_depth = torch.sqrt(torch.sum((start - centre.unsqueeze(1)) ** 2, dim=-1))
    span = torch.zeros_like(dirs[:, :, :1]) + offset * 2
    fine = stratified_samples(n_fine, span, rand)
    fine = fine + start_depth.unsqueeze(-1).unsqueeze(-1)
    coarse = stratified_samples(n_coarse, depths.unsqueeze(-1), rand)
    dists, _ = torch.sort(torch.cat([fine, coarse], -2), -2)
    pts = centre.unsqueeze(1).unsqueeze(1) + dists * dirs.unsqueeze(2)
    return pts, dists


def sdf_ray_loss(rays, pred_sdf, pts, centre, n_valid, truncated_dist):
    """compute_sdf_loss (render_utils.py:508-549): L1 between the decoded SDF and the signed distance to the
    nearest valid neighbouring surface point, on samples in front of / just behind the surface."""
    gt_depth = torch.sqrt(torch.sum((rays["gt_pts"] - centre.unsqueeze(1)) ** 2, dim=-1)).unsqueeze(-1)
    depth = torch.sqrt(torch.sum((pts - centre.unsqueeze(1).unsqueeze(1)) ** 2, dim=-1))
    gt_sdf = torch.clip(gt_depth - depth, min=-truncated_dist, max=truncated_dist)
    valid = gt_sdf > max(-truncated_dist * 0.5, -0.05)
    d = torch.sqrt(torch.sum((rays["neighbor_pts"].unsqueeze(2) - pts.unsqueeze(3)) ** 2, dim=-1))
    nm = rays["neighbor_masks"].unsqueeze(2).repeat(1, 1, pts.shape[2], 1)
    d = torch.where(nm.bool(), d, torch.ones_like(d) * 10000)
    nearest = torch.min(d, dim=-1)[0]
    sign = torch.where(gt_sdf > 0, torch.ones_like(gt_sdf), torch.ones_like(gt_sdf) * -1)
    target = torch.clip(nearest * sign, min=-truncated_dist, max=truncated_dist)
    l1 = F.l1_loss(pred_sdf, target, reduction="none") * valid
    return (l1 * rays["mask"].unsqueeze(-1)).sum() / n_valid


def calculate_loss(volume, rays, sd, truncated_units, truncated_dist, ray_max_dist, sdf_delta=None, rand=None):
    """calculate_loss + render_with_rays (render_utils.py:461-505, 551-590) -> (loss dict, sampled pts)."""
    rand = rand or (lambda *shape: torch.rand(*shape))
    n_valid = torch.sum(rays["mask"]) + 1e-4
    dirs, centre = camera_rays(rays["uv"], rays["T_wc"], rays["intr_mat"])
    gt_depth = torch.sqrt(torch.sum((rays["gt_pts"] - centre.unsqueeze(1)) ** 2, dim=-1))
    pts, _ = hierarchical_samples(truncated_units * 2, int(ray_max_dist * 5), gt_depth, rays["gt_pts"], dirs,
                                  centre, truncated_dist, rand)
    coords = (pts - volume.min_coords) / volume.voxel_size
    volume.count_optim(get_neighbors(coords))
    pred = volume.decode_pts(pts, sd, sdf_delta=sdf_delta)[..., 0]
    return {"depth_bce_loss": sdf_ray_loss(rays, pred, pts, centre, n_valid, truncated_dist)}, pts


# --------------------------------------------------------------------------- #
# per-voxel marching cubes.  PARITY UNPINNED: the reference calls
# skimage.measure.marching_cubes (scikit-image 0.18.3, Lewiner et al. 2003) on every active voxel's 3x3x3
# lattice (sparse_volume.py:740-751); scikit-image is absent here and its 33-case tables cannot be restated
# from memory.  What every marching-cubes variant shares is restated: one vertex per sign-changing lattice
# edge at the linear-interpolation point, triangles spanning exactly those vertices inside each cell, the
# per-voxel gate ``max > level and min < level``, and the coordinate chain of :749-756.  The triangulation of
# a cell (which vertices are joined) follows the face rule documented in bnv_fusion_amd/mc_tables.py and
# may differ from Lewiner's in ambiguous configurations.
# --------------------------------------------------------------------------- #
_MC_CORNERS = [((c >> 2) & 1, (c >> 1) & 1, c & 1) for c in range(8)]
_MC_EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]


def _mc_cell_loops(inside):
    """Closed loops of crossed edges of one cube: per face join crossed edges pairwise (4 crossed: around
    each inside corner), then walk the resulting degree-2 graph."""
    eid = {e: i for i, e in enumerate(_MC_EDGES)}
    link = {}
    for axis in range(3):
        for side in (0, 1):
            ring = []
            for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [du, dv]
                p.insert(axis, side)
                ring.append(4 * p[0] + 2 * p[1] + p[2])
            fe = [eid[tuple(sorted((ring[k], ring[(k + 1) % 4])))] for k in range(4)]
            cr = [k for k in range(4) if inside[ring[k]] != inside[ring[(k + 1) % 4]]]
            if len(cr) == 2:
                segs = [(fe[cr[0]], fe[cr[1]])]
            elif len(cr) == 4:
                segs = [(fe[k - 1], fe[k]) for k in range(4) if inside[ring[k]]]
            else:
                segs = []
            for a, b in segs:
                link.setdefault(a, []).append(b)
                link.setdefault(b, []).append(a)
    loops, todo = [], set(link)
    while todo:
        start = min(todo)
        loop, prev, cur = [start], -1, start
        todo.discard(start)
        while True:
            cand = [n for n in link[cur] if n != prev] or link[cur]
            n = cand[0]
            if n == start:
                break
            loop.append(n)
            todo.discard(n)
            prev, cur = cur, n
        loops.append(loop)
    return loops


def marching_cubes_voxels(sdf, origins, voxel_size, min_coords, level=0.0):
    """sdf [n, 3, 3, 3] (lattice {-.5, 0, .5}^3 of every voxel), origins [n, 3] int -> (vertices [3T, 3]
    float32 world coordinates, faces [T, 3]) as a triangle soup, voxels / cells / loops in order."""
    sdf = np.asarray(sdf, dtype=np.float32)
    origins = np.asarray(origins)
    mn = np.asarray(min_coords, dtype=np.float32)
    verts = []
    for v in range(len(sdf)):
        s = sdf[v]
        if not (s.max() > level and s.min() < level):            # sparse_volume.py:742
            continue
        for cx in range(2):
            for cy in range(2):
                for cz in range(2):
                    val = [s[cx + dx, cy + dy, cz + dz] for dx, dy, dz in _MC_CORNERS]
                    inside = [x < level for x in val]
                    if all(inside) or not any(inside):
                        continue
                    pos = {}
                    for e, (a, b) in enumerate(_MC_EDGES):
                        if inside[a] != inside[b]:
                            t = np.float32(level - val[a]) / np.float32(val[b] - val[a])
                            pa = np.array(_MC_CORNERS[a], np.float32) + np.array([cx, cy, cz], np.float32)
                            pb = np.array(_MC_CORNERS[b], np.float32) + np.array([cx, cy, cz], np.float32)
                            p = (pa + t * (pb - pa)) * np.float32(0.5)            # spacing 0.5 (:719)
                            p = p + (origins[v].astype(np.float32) - np.float32(0.5))   # :749
                            pos[e] = p * np.float32(voxel_size) + mn              # :756
                    for loop in _mc_cell_loops(inside):
                        mid = {e: (np.array(_MC_CORNERS[_MC_EDGES[e][0]], float)
                                   + np.array(_MC_CORNERS[_MC_EDGES[e][1]], float)) / 2 for e in loop}
                        nrm = sum(np.cross(mid[loop[k]], mid[loop[(k + 1) % len(loop)]]) for k in range(len(loop)))
                        g = sum((np.array(_MC_CORNERS[_MC_EDGES[e][1]], float) - np.array(_MC_CORNERS[_MC_EDGES[e][0]], float))
                                * (1 if inside[_MC_EDGES[e][0]] else -1) for e in loop)
                        if np.dot(nrm, g) < 0:
                            loop = loop[::-1]
                        for k in range(1, len(loop) - 1):
                            verts += [pos[loop[0]], pos[loop[k]], pos[loop[k + 1]]]
    if not verts:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64)
    verts = np.stack(verts).astype(np.float32)
    return verts, np.arange(len(verts), dtype=np.int64).reshape(-1, 3)


def _mc_lattice_edge(pa, axis):
    """Id of the lattice edge pa -> pa + e_axis of a 3x3x3 node lattice (18 per axis; csrc/mesh.hip)."""
    if axis == 0:
        return pa[0] * 9 + pa[1] * 3 + pa[2]
    if axis == 1:
        return 18 + pa[0] * 6 + pa[1] * 3 + pa[2]
    return 36 + pa[0] * 6 + pa[1] * 2 + pa[2]


def marching_cubes_voxel_indexed(s, level=0.0):
    """One voxel's 3x3x3 lattice -> (verts [v, 3] in lattice-index units, faces [t, 3] local indices): what
    skimage.measure.marching_cubes(sdf[j], level) returns for it up to (i) the order of the vertices -- here ascending
    lattice-edge id --, (ii) the order / starting corner / diagonal choice of the triangles of a cell and (iii) the
    triangulation inside AMBIGUOUS cells (Lewiner's MC33 topology tests; scikit-image absent -> unpinned).  Shared
    with every variant: one vertex per sign-changing lattice edge at the linear-interpolation point, and in a
    non-ambiguous cell the polygon(s) the triangles span."""
    s = np.asarray(s, dtype=np.float32)
    tri_edges, pos = [], {}
    for cx in range(2):
        for cy in range(2):
            for cz in range(2):
                cc = (cx, cy, cz)
                val = [s[cx + dx, cy + dy, cz + dz] for dx, dy, dz in _MC_CORNERS]
                inside = [x < level for x in val]
                if all(inside) or not any(inside):
                    continue
                lat = {}
                for e, (a, b) in enumerate(_MC_EDGES):
                    if inside[a] != inside[b]:
                        pa = [cc[d] + _MC_CORNERS[a][d] for d in range(3)]
                        pb = [cc[d] + _MC_CORNERS[b][d] for d in range(3)]
                        axis = [d for d in range(3) if pa[d] != pb[d]][0]
                        lid = _mc_lattice_edge(pa, axis)
                        lat[e] = lid
                        t = np.float32(level - val[a]) / np.float32(val[b] - val[a])
                        pos[lid] = np.array(pa, np.float32) + t * (np.array(pb, np.float32) - np.array(pa, np.float32))
                for loop in _mc_cell_loops(inside):
                    mid = {e: (np.array(_MC_CORNERS[_MC_EDGES[e][0]], float)
                               + np.array(_MC_CORNERS[_MC_EDGES[e][1]], float)) / 2 for e in loop}
                    nrm = sum(np.cross(mid[loop[k]], mid[loop[(k + 1) % len(loop)]]) for k in range(len(loop)))
                    g = sum((np.array(_MC_CORNERS[_MC_EDGES[e][1]], float) - np.array(_MC_CORNERS[_MC_EDGES[e][0]], float))
                            * (1 if inside[_MC_EDGES[e][0]] else -1) for e in loop)
                    if np.dot(nrm, g) < 0:
                        loop = loop[::-1]
                    for k in range(1, len(loop) - 1):
                        tri_edges.append((lat[loop[0]], lat[loop[k]], lat[loop[k + 1]]))
    ids = sorted(pos)
    rank = {e: i for i, e in enumerate(ids)}
    verts = np.stack([pos[e] for e in ids]).astype(np.float32) if ids else np.zeros((0, 3), np.float32)
    faces = np.array([[rank[e] for e in t] for t in tri_edges], dtype=np.int64).reshape(-1, 3)
    return verts, faces


def meshlize_concat(sdf, origins, voxel_size, min_coords, level=0.0):
    """The mesh assembly of SparseVolume.meshlize (sparse_volume.py:740-756) around the per-voxel mesher above:
    gate, spacing 0.5, ``verts += origin - 0.5``, ``faces + last_face_id``, ``last_face_id += max(faces) + 1``,
    ``* voxel_size + min_coords``.  -> (vertices [V, 3] f32, faces [T, 3] i64)."""
    sdf = np.asarray(sdf, dtype=np.float32).reshape(-1, 3, 3, 3)
    origins = np.asarray(origins)
    mn = np.asarray(min_coords, dtype=np.float32)
    all_v, all_f, last_face_id = [], [], 0
    for j in range(len(sdf)):
        if np.max(sdf[j]) > level and np.min(sdf[j]) < level:                      # :742
            verts, faces = marching_cubes_voxel_indexed(sdf[j], level)
            verts = verts * np.float32(0.5)                                          # spacing (:719, :746)
            verts = verts + (origins[j].astype(np.float32) - np.float32(0.5))       # :749
            all_v.append(verts)
            all_f.append(faces + last_face_id)                                      # :751
            last_face_id += int(np.max(faces)) + 1                                  # :752
    if not all_v:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64)
    v = np.concatenate(all_v, 0) * np.float32(voxel_size) + mn                      # :756
    return v.astype(np.float32), np.concatenate(all_f, 0)


def synthetic_depth(t, H=480, W=640, seed=0):
    """depth(u,v) = 1.5 + 0.2 sin(u/40) cos(v/30) + N(0, 0.002) m,
    quantised to uint16 millimetres as the datasets store it (common.py:93)."""
    rng = np.random.default_rng(seed + 1000 * t)
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    d = 1.5 + 0.2 * np.sin(u / 40.0) * np.cos(v / 30.0) + rng.normal(0.0, 0.002, size=(H, W))
    return np.round(d * 1000.0).astype(np.uint16).astype(np.float64) / 1000.0


def synthetic_pose(t):
    """T_wc(t) = translate(0, 0, -1.5) . R_y(0.5 deg * t)."""
    a = math.radians(0.5 * t)
    T = np.eye(4)
    T[:3, :3] = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    T[:3, 3] = [0.0, 0.0, -1.5]
    return T


SYNTHETIC_INTRINSICS = np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1.0]])


def synthetic_frame(t, H=480, W=640, seed=0):
    """One benchmark frame -> input_pts [1, N, 6] float32 (float64 -> .float(), run_e2e.py:249)."""
    pts = depth_to_input_pts(synthetic_depth(t, H, W, seed), SYNTHETIC_INTRINSICS, synthetic_pose(t))
    return torch.from_numpy(pts).float().unsqueeze(0)
