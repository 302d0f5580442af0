import sys, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic
from bnv_fusion_amd.neural_map import frame_input_pts
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
vol = bnv.SparseVolume(8, voxel, np.array([dims]*3), 8, device="cuda:0")
for t in (0, 10, 35):
    fr = {"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)}
    pts = frame_input_pts(fr)
    f, c, ids, g, cnt, cap = model.encode_pointcloud_async(pts, vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size)
    h = cnt.cpu()
    n_valid, n_unique, n_out = int(h[0]), int(h[1]), int(h[2])
    kept_pairs = int(c[:n_out].sum())
    print(f"frame {t}: valid points {n_valid}, pairs {8*n_valid}, touched voxels U={n_unique}, kept U'={n_out}, pairs in kept voxels {kept_pairs} = {kept_pairs/(8*n_valid):.3f}")
