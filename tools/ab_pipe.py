"""A/B of the lattice-table kernels: k_decode<LATTICE,1> (lattice_pipe=0: 32x32x16 MFMA) and k_lattice_table_x (=1:
16x16x32 MFMA, same arithmetic in another summation grouping): differences of the decoded SDF and interleaved
kernel timings.  Usage: ab_pipe.py [mlp_mode]"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, _lib
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<20, device="cuda:0", tsdf=True)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(60)]
for t in range(30): nm.integrate(frames[t])
lib = _lib.load()
if len(sys.argv) > 1:
    bnv.set_mlp_mode(int(sys.argv[1]))
coords = nm.integrate(frames[30])
outs = {}
for opt in (0, 1):
    lib.bnv_set_option(b"lattice_pipe", opt)
    outs[opt] = nm.volume.decode_lattice(coords, model.nerf, None, query_tensor=False).clone()
print("pipe 1 vs 0: bitwise equal:", torch.equal(outs[0], outs[1]), "max diff", float((outs[0]-outs[1]).abs().max()),
      "mask decisions equal:", bool(((outs[0] == voxel) == (outs[1] == voxel)).all()),
      "live", float((outs[1] != voxel).float().mean()))
# small / ragged sizes
for n in (1, 5, 129, 1000):
    a = {}
    for opt in (0, 1):
        lib.bnv_set_option(b"lattice_pipe", opt)
        a[opt] = nm.volume.decode_lattice(coords[:n], model.nerf, None, query_tensor=False).clone()
    print(n, float((a[0] - a[1]).abs().max()))
res = {0: [], 1: []}
for rnd in range(4):
    for opt in (0, 1):
        lib.bnv_set_option(b"lattice_pipe", opt)
        nm.fuse_and_decode(frames[30]); torch.cuda.synchronize()
        lib.bnv_profile_enable(1)
        for t in range(31, 51): nm.fuse_and_decode(frames[t])
        torch.cuda.synchronize()
        ms=(C.c_double*4)(); n=(C.c_int64*4)(); lib.bnv_profile_read(ms,n); lib.bnv_profile_enable(0)
        res[opt].append(ms[1]/n[1])
print("lattice MLP kernel ms  pipe=0:", ["%.3f"%x for x in res[0]], " pipe=1:", ["%.3f"%x for x in res[1]])
lib.bnv_set_option(b"lattice_pipe", 1)
