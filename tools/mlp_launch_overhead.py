"""Fixed cost per launch of the two persistent MLP kernels: duration (HIP events on the launch stream) against the
amount of work, from an empty launch to the full 640x480 frame.  At world 8 a rank runs 1/8-size launches every
frame, so the intercept -- dispatch, weight staging, pipeline fill and drain, tail -- is paid in full.

    python tools/mlp_launch_overhead.py [--reps 40]
"""
import argparse, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--mode", type=int, default=1)
args = ap.parse_args()
dev = "cuda:0"
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device=dev, voxel_size=voxel)
bnv.set_mlp_mode(args.mode)
nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=1 << 21, device=dev)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).to(dev), "intr_mat": synthetic.intrinsics(),
           "T_wc": synthetic.pose(t)} for t in range(34)]
for f in frames[:30]:
    nm.integrate(f)
coords, sdf = nm.fuse_and_decode(frames[30])
lib = _lib.load()
vol = nm.volume


def timed(fn, kind):
    fn()
    torch.cuda.synchronize()
    lib.bnv_profile_enable(1)
    for _ in range(args.reps):
        fn()
    ms, cnt = (C.c_double * 4)(), (C.c_int64 * 4)()
    lib.bnv_profile_read(ms, cnt)
    lib.bnv_profile_enable(0)
    return 1e3 * ms[kind] / max(cnt[kind], 1)


# ---- lattice-table kernel: the frame's own entry list, truncated to N entries ------------------------------------
n = int(coords.shape[0])
ws = vol._lattice_ws
off_cnt = int(lib.bnv_decode_lattice_count_offset(vol._row_capacity))
n_list = ws[off_cnt: off_cnt + 16].view(torch.int32)
full = int(n_list[1])
print(f"lattice table kernel (mode {args.mode}): the frame lists {full} entries (tiles of 128 evaluations, {lib.bnv_num_compute_units()} workgroups)")
f_, w_ = vol._features, vol._weights
xs, ys = [], []
for frac in (0.0, 1 / 64, 1 / 16, 1 / 8, 1 / 4, 1 / 2, 1.0):
    N = int(full * frac)
    n_list[1] = N

    def run():
        _lib.check(lib.bnv_lattice_table(C.byref(vol._struct()), C.byref(vol._grid), _lib.ptr(f_),
                                         _lib.ptr(model.nerf.sdf_pack), n, 1, _lib.ptr(ws), ws.numel(),
                                         _lib.stream_ptr()), "table")
    us = timed(run, 1)
    tiles = -(-N // 128)
    xs.append(tiles / lib.bnv_num_compute_units()); ys.append(us)
    print(f"  {N:8d} entries = {tiles:6d} tiles = {xs[-1]:6.2f} per workgroup: {us:8.1f} us")
b, a = np.polyfit(xs[3:], ys[3:], 1)
print(f"  fit over the upper four: {a:.1f} us + {b:.2f} us per tile and workgroup")
n_list[1] = full

# ---- point encoder: the first N points of the frame -------------------------------------------------------------
from bnv_fusion_amd.frontend import depth_to_input_pts
pts = depth_to_input_pts(frames[31]["depth"], frames[31]["intr_mat"], frames[31]["T_wc"], max_depth=3.0, compact=False)[0]
P = int(pts.shape[1])
print(f"point encoder (mode {args.mode}): {P} points, tiles of 32 pairs, 8 waves per workgroup")
xs, ys = [], []
for frac in (1 / 1024, 1 / 64, 1 / 16, 1 / 8, 1 / 4, 1 / 2, 1.0):
    N = max(int(P * frac) // 32 * 32, 32)
    sub = pts[:, :N].contiguous()

    def run():
        model.encode_pointcloud_async(sub, vol.n_xyz, vol.min_coords, vol.max_coords, voxel)
    us = timed(run, 0)
    tiles = N // 32 * 8
    xs.append(tiles / (8 * lib.bnv_num_compute_units())); ys.append(us)
    print(f"  {N:8d} points = {tiles:6d} tiles = {xs[-1]:6.2f} per wave: {us:8.1f} us")
b, a = np.polyfit(xs[3:], ys[3:], 1)
print(f"  fit over the upper four: {a:.1f} us + {b:.2f} us per tile and wave")
