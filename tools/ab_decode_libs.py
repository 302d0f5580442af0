"""Interleaved A/B of the lattice-table kernel between library builds: python tools/ab_decode_libs.py lib1.so lib2.so ...
(each library in its own child process; 3 rounds; prints the kernel time over 20 frames after a 30-frame pre-roll)."""
import subprocess, sys, os
CHILD = r'''
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, _lib
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<20, device="cuda:0", tsdf=True)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(52)]
for t in range(30): nm.integrate(frames[t])
lib = _lib.load()
nm.fuse_and_decode(frames[30]); torch.cuda.synchronize()
lib.bnv_profile_enable(1)
for t in range(31, 51): nm.fuse_and_decode(frames[t])
torch.cuda.synchronize()
ms=(C.c_double*4)(); n=(C.c_int64*4)(); lib.bnv_profile_read(ms,n)
print("%.4f %.4f" % (ms[1]/n[1], ms[0]/n[0]))
'''
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(5):
    for l in libs:
        env = dict(os.environ, BNV_FUSION_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        res[l].append(out)
for l in libs:
    print(l, "decode_ms enc_ms:", res[l])
