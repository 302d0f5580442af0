import os, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
G = "tests/golden/"
DEV = "cuda:0"
for mode in (1, 3):
    bnv.set_mlp_mode(mode)
    model = bnv.load_pretrained(device=DEV, voxel_size=0.02)
    z = np.load(G + "encode_64.npz")
    vol = bnv.SparseVolume(8, float(z["voxel_size"]), z["dims"], 8, device=DEV)
    f, c, ids, g, n = model.encode_pointcloud(torch.from_numpy(z["input_pts"]).to(DEV), vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size, return_dense=False)
    ferr = np.abs(f.cpu().numpy() - z["feats"]).max()
    seq = np.load(G + "sequence_64.npz"); dec = np.load(G + "decode_64.npz")
    vol = bnv.SparseVolume(8, float(seq["voxel_size"]), seq["dims"], 8, device=DEV)
    for fr in seq["frames"]:
        f, c, _, g, n = model.encode_pointcloud(torch.from_numpy(fr).to(DEV), vol.n_xyz, vol.min_coords, vol.max_coords, vol.voxel_size, return_dense=False)
        model._integrate(vol, g, f, c)
    vol.to_tensor()
    e = {}
    for key, coords in (("lattice_qt", dec["lattice_coords"]), ("random_qt", dec["random_coords"])):
        out = vol.decode_pts(torch.from_numpy(coords).to(DEV), model.nerf, None, is_coords=True).cpu().numpy()
        e[key] = float(np.abs(out - dec[key]).max())
    print("mode", mode, "encoder feats max err %.2e" % ferr, "end-to-end SDF max err", {k: "%.2e" % v for k, v in e.items()})
bnv.set_mlp_mode(1)
