"""200 lattice decodes of one frame's voxels on a fixed volume (for rocprofv3 --kernel-trace --stats A/B runs of the
decode-side kernels: `BNV_FUSION_LIB=<other .so>` selects the library).
    python tools/decode_loop.py [sweep|bench] [fp32|tcnn]"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import sequence, synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "sweep"
tc = (sys.argv[2] if len(sys.argv) > 2 else "tcnn") == "tcnn"
if which == "sweep":
    dims, voxel, scale = sequence.DIMS[512]
    frames = list(sequence.sweep_frames(range(0, 40), scale=scale, device="cuda:0"))
else:
    dims, voxel = synthetic.GRID_DIMS[256]
    frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(),
               "T_wc": synthetic.pose(t)} for t in range(40)]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel, tiny_cuda=tc)
nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=1 << 21, device="cuda:0")
for f in frames:
    c = nm.integrate(f)
print("voxels", tuple(c.shape))
for rep in range(200):
    out = nm.volume.decode_lattice(c, model.nerf, None, query_tensor=False)
torch.cuda.synchronize()
print("evals", int(nm.volume.last_lattice_evals()), "checksum", sequence.checksum(out))
