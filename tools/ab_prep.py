"""A/B of the NeuralMap frame pipeline: upsert + TSDF + first decode stage on a side stream beside the previous
frame's SDF-MLP kernel (overlap_prep) vs everything behind it on the main stream; bitwise equality + frames/s."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
from bnv_fusion_amd import synthetic
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(80)]
outs = {}
for rnd in range(3):
    for prep in (False, True):
        nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<20, device="cuda:0", tsdf=True)
        nm.overlap_prep = prep
        for f in frames[:30]: nm.fuse_and_decode_async(f, decode=False)
        hs = [nm.fuse_and_decode_async(f) for f in frames[30:35]]
        [h.result() for h in hs]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pending, res = None, []
        for f in frames[35:75]:
            h = nm.fuse_and_decode_async(f)
            if pending is not None: res.append(pending.result())
            pending = h
        res.append(pending.result())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"overlap_prep={prep}: {40/dt:.1f} frames/s ({1e3*dt/40:.3f} ms/frame)")
        if rnd == 0:
            outs[prep] = [(c.clone(), s.clone()) for c, s in res]
            nm.volume.to_tensor(); outs[(prep, "vol")] = (nm.volume.features.clone(), nm.volume.weights.clone(), nm.tsdf_vol.tsdf.clone())
print("bitwise equal outputs:", all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(outs[False], outs[True])),
      "volumes:", all(torch.equal(a, b) for a, b in zip(outs[(False, "vol")], outs[(True, "vol")])))
