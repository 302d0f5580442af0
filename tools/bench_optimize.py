"""Throughput of the global optimiser (run_e2e.py:111-162) at the reference's configuration: 5000 rays per
step, 1000 rays per backward, 20 fine + 15 coarse samples per ray, 256^3 volume at 1 cm.  Prints the
"speed on global fusion" figure the reference logs (run_e2e.py:289: optimisation steps per second) and the
time of the decode_pts forward / backward kernels."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bnv_fusion_amd as bnv  # noqa: E402
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
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
print("volume rows", nm.volume.num_rows())
for n_iters in (5, 40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = nm.optimize(n_iters=n_iters, last_frame=-1, generator=None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n_iters} steps: {dt / n_iters * 1e3:.2f} ms/step = {n_iters / dt:.1f} steps/s; loss {float(hist[0]):.4f} -> {float(hist[-1]):.4f}")

# kernel-only timing of one split (1000 rays x 35 samples)
vol = nm.volume
vol.to_tensor()
vol.features = torch.nn.Parameter(vol.features)
f = nm.frames[3]
rays = optimize.sample_key_frame(f["depth"].float() / 1000.0, f["intr_mat"], f["T_wc"], 1000, 3)
out = optimize.render_with_rays(vol, rays, model.nerf, None, nm.truncated_units, nm.truncated_dist, 3)
pts = out["pts_on_rays"].detach()
live = float((out["sdf_on_rays"].detach() != voxel).float().mean())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for _ in range(3):
    ev[0].record()
    sdf = vol.decode_pts(pts, model.nerf, None)
    ev[1].record()
    sdf.sum().backward()
    ev[2].record()
torch.cuda.synchronize()
print(f"35,000 queries ({live:.2f} live): forward {ev[0].elapsed_time(ev[1]):.3f} ms, backward {ev[1].elapsed_time(ev[2]):.3f} ms")

# whole-volume mesh extraction (run_e2e.py:164-167): lattice decode of every active voxel + per-voxel marching cubes
vol.features = vol.features.detach()
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = nm.extract_mesh()
    torch.cuda.synchronize()
    print(f"extract_mesh: {1e3 * (time.perf_counter() - t0):.2f} ms, {vol.num_rows()} active voxels, {len(m.vertices)} vertices, {len(m.faces)} faces")
