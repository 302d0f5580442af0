import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, _lib
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
vol = bnv.SparseVolume(8, voxel, np.array([dims]*3), 8, device="cuda:0")
pts = torch.from_numpy(synthetic.frame(0)).cuda()
lib = _lib.load()
for mode in (1, 0):
    bnv.set_mlp_mode(mode)
    for _ in range(3): model.encode_pointcloud(pts, vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size, return_dense=False)
    lib.bnv_profile_enable(1)
    for _ in range(10): model.encode_pointcloud(pts, vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size, return_dense=False)
    ms=(C.c_double*4)(); n=(C.c_int64*4)(); lib.bnv_profile_read(ms,n); lib.bnv_profile_enable(0)
    print("mode", mode, "pointnet kernel avg ms", ms[0]/n[0])
