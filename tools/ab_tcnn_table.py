"""tiny-cuda-nn block encoder: one LDS table per wave and 8 x 4-pixel block against one per workgroup and 16 x 16
patch -- time of the whole encode (HIP events on one stream) and bit-identity of the outputs."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, _lib
dev = "cuda:0"
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device=dev, voxel_size=voxel, tiny_cuda=True)
nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=1 << 20, device=dev)
v = nm.volume
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).to(dev), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(16)]
lib = _lib.load()
ref = None
for per in (0, 1, 0, 1):
    assert lib.bnv_set_option(b"tcnn_shared_table", per) == 0
    outs = []
    t = []
    for rep in range(6):
        for f in frames:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            feats, pcounts, flat_ids, grid_ids, counters, cap, _ = model.encode_depth_async(
                f["depth"], f["intr_mat"], f["T_wc"], nm.max_depth, v.n_xyz, v.min_coords, v.max_coords, v.voxel_size)
            b.record()
            torch.cuda.synchronize()
            t.append(a.elapsed_time(b))
            if rep == 0:
                n = int(counters[2])
                outs.append((feats[:n].clone(), pcounts[:n].clone(), grid_ids[:n].clone()))
    if ref is None:
        ref = outs
    same = all(torch.equal(x, y) for o, r in zip(outs, ref) for x, y in zip(o, r))
    print(f"shared table {per}: whole encode {1e3 * np.median(t):.1f} us (median of {len(t)}), outputs equal to the per-wave tables': {same}")
