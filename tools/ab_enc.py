"""A/B of the point-encoder variants (bnv_set_option encoder_overlap 0/1): bitwise equality + timings."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
from bnv_fusion_amd import synthetic, _lib
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<20, device="cuda:0", tsdf=True)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(60)]
lib = _lib.load()
from bnv_fusion_amd.neural_map import frame_input_pts
pts = frame_input_pts(frames[3])
v = nm.volume
outs = {}
for opt in (0, 1):
    lib.bnv_set_option(b"encoder_overlap", opt)
    f, c, ids, g, n = model.encode_pointcloud(pts, v.n_xyz, v.min_coords, v.max_coords, v.voxel_size, return_dense=False)
    outs[opt] = (f.clone(), c.clone(), ids.clone())
print("feats bitwise equal:", torch.equal(outs[0][0], outs[1][0]), "max diff", float((outs[0][0]-outs[1][0]).abs().max()),
      "ids equal", torch.equal(outs[0][2], outs[1][2]), "counts equal", torch.equal(outs[0][1], outs[1][1]))
res = {0: [], 1: []}
for rnd in range(4):
    for opt in (0, 1):
        lib.bnv_set_option(b"encoder_overlap", opt)
        nm.integrate(frames[30]); torch.cuda.synchronize()
        lib.bnv_profile_enable(1)
        for t in range(31, 51): nm.integrate(frames[t])
        torch.cuda.synchronize()
        ms=(C.c_double*4)(); n=(C.c_int64*4)(); lib.bnv_profile_read(ms,n); lib.bnv_profile_enable(0)
        res[opt].append(ms[0]/n[0])
print("encoder kernel ms  overlap=0:", ["%.3f"%x for x in res[0]], " overlap=1:", ["%.3f"%x for x in res[1]])
