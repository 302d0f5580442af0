"""The optimiser loop alone (NeuralMap.optimize at the reference's configuration: 5,000 rays, splits of 1,000) for
`rocprofv3 --kernel-trace --stats` and for a host-side breakdown: steps/s, the time of the step's phases with a device
synchronisation behind each (what the GPU work of a phase takes when nothing overlaps), and the host's enqueue time."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bnv_fusion_amd as bnv  # noqa: E402
bnv.configure_runtime()
from bnv_fusion_amd import optimize, synthetic  # noqa: E402

DEV = "cuda:0"
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=2_000_000, device=DEV, tsdf=True)
for t in range(40):
    fr = {"depth": torch.from_numpy(synthetic.depth_u16(t)).to(DEV), "intr_mat": synthetic.intrinsics(),
          "T_wc": synthetic.pose(t)}
    nm.integrate(fr)
    nm.frames.append(fr)
torch.cuda.synchronize()
gen = torch.Generator(device=DEV).manual_seed(0)
nm.optimize(n_iters=5, last_frame=-1, generator=gen)
for n_iters in (40, 200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = nm.optimize(n_iters=n_iters, last_frame=-1, generator=gen)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n_iters} steps: {dt / n_iters * 1e3:.3f} ms/step = {n_iters / dt:.1f} steps/s")

# phases of one step, synchronised
vol = nm.volume
vol.to_tensor()
vol.features = torch.nn.Parameter(vol.features)
opt = torch.optim.Adam([vol.features], lr=1e-3, fused=True)      # as optimize.optimize_volume
vol.features.grad = torch.zeros_like(vol.features)
f = nm.frames[3]
pts_cache = optimize.key_frame_points(f["depth"].float() / 1000.0, f["intr_mat"], f["T_wc"], 3)
delta = nm.prepare_tsdf_volume()
acc = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    a = acc.setdefault(name, [0.0, 0.0])
    a[0] += t1 - t0
    a[1] += t2 - t0
    return out


for it in range(30):
    if it == 10:
        acc.clear()
    rays = timed("sample_key_frame", lambda: optimize.sample_key_frame(None, None, None, 5000, 3, gen, points=pts_cache))
    timed("zero_grad", lambda: opt.zero_grad(set_to_none=False))
    timed("ray_batch_step", lambda: optimize.ray_batch_step(vol, rays, model.nerf, nm.truncated_units, nm.truncated_dist, 3,
                                                            sdf_delta=delta, generator=gen, grad=vol.features.grad))
    timed("adam", lambda: opt.step())
print("phase: host enqueue ms / synchronised ms (mean of 20)")
for k, (h, s) in acc.items():
    print(f"  {k:18s} {1e3 * h / 20:.3f} / {1e3 * s / 20:.3f}")
vol.features = vol.features.detach()
