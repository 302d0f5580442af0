import sys, os, numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from oracle import bnv_oracle as orc
DEV="cuda:0"
z = np.load("tests/golden/sequence_64.npz")
dims, voxel = z["dims"], float(z["voxel_size"])
tsd = orc.load_weights("bnv_fusion_amd/weights/pointnet_tcnn.npz")
geo = orc.tcnn_geo_forward(tsd["nerf.model.params"])
model = bnv.load_pretrained(device=DEV, voxel_size=voxel, tiny_cuda=True)
nm = bnv.NeuralMap(dims, voxel, model, device=DEV)
for fr in z["frames"]:
    nm.integrate({"input_pts": torch.from_numpy(fr).to(DEV)})
nm.volume.to_tensor()
k = nm.volume.active_coordinates.cpu()
ov2 = orc.OracleSparseVolume(8, voxel, dims, 8)
ov2.insert(k, nm.volume.features.cpu(), nm.volume.weights.cpu(), nm.volume.num_hits.cpu()); ov2.to_tensor()
valid = (nm.volume.weights[:, 0] >= 8).nonzero()[:, 0][:120]
origins = nm.volume.active_coordinates[valid]
ref = ov2.decode_pts(orc.lattice_coords(origins.cpu().numpy()), None, None, is_coords=True, geo=geo)[0, :, :, 0]
L = torch.tensor([[x, y, z] for x in (-.5, 0, .5) for y in (-.5, 0, .5) for z in (-.5, 0, .5)], device=DEV)
coords = origins[:, None, :].float() + L[None]
for it in range(3):
    lat = nm.volume.decode_lattice(origins, model.nerf, query_tensor=True).cpu()
    gen = nm.volume.decode_pts(coords[None], model.nerf, None, is_coords=True)[0, :, :, 0].cpu()
    live = ref != voxel
    print(it, "lat-ref", float((lat-ref).abs().max()), "gen-ref", float((gen-ref).abs().max()), "n live", int(live.sum()),
          "gen bad count", int(((gen-ref).abs() > 1e-4).sum()), "lat bad", int(((lat-ref).abs() > 1e-4).sum()))
bad = ((gen-ref).abs() > 1e-4).nonzero()
print(bad[:10].tolist())
for b,p in bad[:5].tolist(): print(b,p, float(gen[b,p]), float(ref[b,p]), float(lat[b,p]))
