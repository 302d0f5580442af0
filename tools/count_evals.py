"""How many table entries does a frame's lattice decode evaluate?  Same frames, same volume history: the two marking paths
(separate neighbour kernel + k_lattice_mark<false>  |  fused k_lattice_mark<true>) and, for comparison, the sum over the 8
shards of a simulated world (tools/spatial_single_rank.py prints that).  Diagnostic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bnv_fusion_amd as bnv
bnv.configure_runtime()
from bnv_fusion_amd import synthetic, _lib
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)}
          for t in range(60)]
lib = _lib.load()
res = {}
for fused in (0, 1):
    _lib.check(lib.bnv_set_option(b"fused_mark", fused), "fused_mark")
    m = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=1 << 21, device="cuda:0", tsdf=False)
    m.frame_pipe = False
    ev, outs = [], []
    for t, fr in enumerate(frames):
        c, s = m.fuse_and_decode(fr)
        ev.append(int(m.volume.last_lattice_evals()[0]))
        if t >= 50:
            outs.append((c.clone(), s.clone()))
    live = float((outs[-1][1] != voxel).float().mean())
    print(f"fused_mark={fused}: evaluations per frame (frames 40..59) {np.mean(ev[40:]):.0f}, voxels {len(outs[-1][0])}, live fraction {live:.3f}, "
          f"live lattice points {int((outs[-1][1] != voxel).sum())}")
    res[fused] = outs
same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(res[0], res[1]))
print("outputs equal:", same)
_lib.check(lib.bnv_set_option(b"fused_mark", -1), "fused_mark")
