"""Does a small kernel of another stream start beside the persistent MLP kernels (point encoder P, lattice-table
decoder T)?  Enqueue the big kernel on stream A, then a 1 MB fill (4 VGPRs) on stream B, and compare completion times."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<22, device="cuda:0", tsdf=False)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(34)]
for f in frames[:30]: nm.integrate(f)
coords = nm.integrate(frames[30])
from bnv_fusion_amd.neural_map import frame_input_pts
pts = frame_input_pts(frames[31])
v = nm.volume
A, B = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1 << 18, device="cuda:0")
def big_encode():
    nm.pointnet.encode_pointcloud_async(pts, v.n_xyz, v.min_coords, v.max_coords, v.voxel_size)
def big_decode():
    v.decode_lattice(coords, model.nerf, None, query_tensor=False)
for name, big in (("encoder P", big_encode), ("decoder T", big_decode)):
    for rep in range(3):
        torch.cuda.synchronize()
        e0, eA, eB = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.cuda.stream(A):
            A.wait_event(e0); big(); eA.record()
        time.sleep(0.0003)                  # the big kernel is running by now
        with torch.cuda.stream(B):
            B.wait_event(e0); x.fill_(1.0); eB.record()
        torch.cuda.synchronize()
        print(f"{name}: big kernel chain done after {e0.elapsed_time(eA):.3f} ms, the small fill on the other stream after {e0.elapsed_time(eB):.3f} ms")
