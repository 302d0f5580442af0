"""Times bnv_lattice_mark alone (HIP events) on a bench-like frame; run once per library build:
BNV_FUSION_LIB=tools/libbnv_mark_<threads>_<chunks>_<scan>.so python tools/ab_mark.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, _lib
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=1 << 22, device="cuda:0", tsdf=False)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(34)]
for f in frames[:33]:
    nm.integrate(f)
coords, sdf = nm.fuse_and_decode(frames[33])
vol, lib = nm.volume, _lib.load()
n = int(coords.shape[0])
ws = vol._lattice_ws
args = (C.byref(vol._struct()), C.byref(vol._grid))
ts = []
for rep in range(12):
    vol._lattice_epoch += 1
    _lib.check(lib.bnv_lattice_neighbors(*args, _lib.ptr(vol._weights), vol._row_capacity, _lib.ptr(coords.contiguous()), n, None,
                                         None, 0, _lib.ptr(ws), ws.numel(), vol._lattice_epoch, _lib.stream_ptr()), "nbr")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    _lib.check(lib.bnv_lattice_mark(args[0], n, None, _lib.ptr(ws), ws.numel(), vol._lattice_epoch, _lib.stream_ptr()), "mark")
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
    off = int(lib.bnv_decode_lattice_count_offset(vol._row_capacity))
    cnt = int(ws[off + 4: off + 8].view(torch.int32).item())
    # clean need_mask as the table kernel would
    _lib.check(lib.bnv_lattice_table(args[0], args[1], _lib.ptr(vol._features), _lib.ptr(model.nerf.sdf_pack), n, 1, _lib.ptr(ws),
                                     ws.numel(), _lib.stream_ptr()), "table")
print(f"{os.environ.get('BNV_FUSION_LIB', 'product')}: mark (memset + kernel) min {min(ts[2:])*1e3:.1f} us, median {sorted(ts[2:])[len(ts[2:])//2]*1e3:.1f} us; entries {cnt}, voxels {n}")
