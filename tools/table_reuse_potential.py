"""How many of the table rows a frame's decode needs were NOT updated by that frame's integrate (their SDF-MLP table
entries could be carried over from an earlier frame)?  Row-level estimate on the bench workload."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic
dev = "cuda:0"
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device=dev, voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=1 << 20, device=dev, tsdf=False)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).to(dev), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(60)]
for f in frames[:30]:
    nm.integrate(f)
off = torch.tensor([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)], device=dev)
prev_needed = None
for t in range(30, 60):
    coords = nm.integrate(frames[t])
    nb = torch.unique((coords[:, None, :] + off[None]).reshape(-1, 3), dim=0)
    _, w, _ = nm.volume.query(nb)
    usable = nb[w[:, 0] >= 8]
    key = lambda c: (c[:, 0] * 4096 + c[:, 1]) * 4096 + c[:, 2]
    ku, kc = key(usable), key(coords)
    fresh = torch.isin(ku, kc)
    msg = f"frame {t}: decoded voxels {len(coords)}, table rows needed {len(ku)}, of which updated this frame {int(fresh.sum())} ({100*float(fresh.float().mean()):.1f} %)"
    if prev_needed is not None:
        carried = (~fresh) & torch.isin(ku, prev_needed)
        msg += f"; not updated AND needed by the previous frame too: {100*float(carried.float().mean()):.1f} %"
    prev_needed = ku
    if t % 3 == 0:
        print(msg)
