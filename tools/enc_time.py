"""Point-encoder kernel time (HIP events on its stream) on one synthetic 640x480 frame at 256^3; run it with
BNV_FUSION_LIB=tools/libbnv_noscatter.so (a build whose scatter keeps 1 of 64 workgroups' atomics) to see what the
scatter atomics cost."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, _lib
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
vol = bnv.SparseVolume(8, voxel, np.array([dims] * 3), 8, device="cuda:0")
lib = _lib.load()
for mode in (1, 0, 3):
    bnv.set_mlp_mode(mode)
    frames = [torch.from_numpy(synthetic.frame(t)).cuda() for t in range(4)]
    for rep in range(3):
        for p in frames:
            model.encode_pointcloud_async(p, vol.n_xyz, vol.min_coords, vol.max_coords, voxel)
    torch.cuda.synchronize()
    lib.bnv_profile_enable(1)
    for rep in range(5):
        for p in frames:
            model.encode_pointcloud_async(p, vol.n_xyz, vol.min_coords, vol.max_coords, voxel)
    torch.cuda.synchronize()
    ms, n = (C.c_double * 4)(), (C.c_int64 * 4)()
    lib.bnv_profile_read(ms, n)
    lib.bnv_profile_enable(0)
    print(f"mode {mode}: pointnet+scatter kernel {ms[0] / max(n[0], 1):.4f} ms over {n[0]} launches ({os.environ.get('BNV_FUSION_LIB', 'product build')})")
