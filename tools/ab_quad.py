"""A/B of the lattice-table kernels: k_lattice_table_h (lattice_quad=0, 32 x 128 per wave) vs k_lattice_table_q (=1,
64 x 64 per wave), in split mode (1) and f16-operand mode (3): bitwise equality of the decoded SDF and interleaved
kernel timings on the bench frames."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
from bnv_fusion_amd import synthetic, _lib
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(60)]
lib = _lib.load()
for mode in (1, 3):
    bnv.set_mlp_mode(mode)
    nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<20, device="cuda:0", tsdf=True)
    for t in range(30): nm.integrate(frames[t])
    coords = nm.integrate(frames[30])
    outs = {}
    for opt in (0, 1):
        lib.bnv_set_option(b"lattice_quad", opt)
        outs[opt] = nm.volume.decode_lattice(coords, model.nerf, None, query_tensor=False).clone()
    print(f"mode {mode}: bitwise equal:", torch.equal(outs[0], outs[1]), "max diff", float((outs[0]-outs[1]).abs().max()),
          "live", float((outs[1] != voxel).float().mean()))
    for n in (1, 5, 129, 1000):
        a = {}
        for opt in (0, 1):
            lib.bnv_set_option(b"lattice_quad", opt)
            a[opt] = nm.volume.decode_lattice(coords[:n], model.nerf, None, query_tensor=False).clone()
        print("  n =", n, torch.equal(a[0], a[1]))
    res = {0: [], 1: []}
    for rnd in range(4):
        for opt in (0, 1):
            lib.bnv_set_option(b"lattice_quad", opt)
            nm.fuse_and_decode(frames[30]); torch.cuda.synchronize()
            lib.bnv_profile_enable(1)
            for t in range(31, 51): nm.fuse_and_decode(frames[t])
            torch.cuda.synchronize()
            ms=(C.c_double*4)(); n=(C.c_int64*4)(); lib.bnv_profile_read(ms,n); lib.bnv_profile_enable(0)
            res[opt].append(ms[1]/n[1])
    print(f"mode {mode}: lattice MLP kernel ms  32x128:", ["%.3f"%x for x in res[0]], " 64x64:", ["%.3f"%x for x in res[1]])
lib.bnv_set_option(b"lattice_quad", 0)
