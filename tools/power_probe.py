"""Are the two MLP kernels bound by cycles or by power?  Same kernel, same instruction stream, same launch sequence:
once with the real weights, once with the packed weights zeroed (MFMA operands, activations and outputs are then
zero; the instruction stream is identical).  A large difference in kernel time = the kernel runs at the power limit.
Also prints the MFMA-only ceiling of the box (bnv_probe_mfma_rate) with random and with zero operands."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, _lib
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
vol = bnv.SparseVolume(8, voxel, np.array([dims] * 3), 8, device="cuda:0")
lib = _lib.load()
frames = [torch.from_numpy(synthetic.frame(t)).cuda() for t in range(4)]
real = model.pointnet_pack.clone()


def timeit(tag, reps=25):
    for rep in range(5):
        for p in frames:
            model.encode_pointcloud_async(p, vol.n_xyz, vol.min_coords, vol.max_coords, voxel)
    torch.cuda.synchronize()
    lib.bnv_profile_enable(1)
    for rep in range(reps):
        for p in frames:
            model.encode_pointcloud_async(p, vol.n_xyz, vol.min_coords, vol.max_coords, voxel)
    torch.cuda.synchronize()
    ms, n = (C.c_double * 4)(), (C.c_int64 * 4)()
    lib.bnv_profile_read(ms, n)
    lib.bnv_profile_enable(0)
    print(f"{tag}: pointnet+scatter kernel {ms[0] / max(n[0], 1):.4f} ms over {n[0]} launches")


for mode in (1, 3):
    bnv.set_mlp_mode(mode)
    for rnd in range(2):
        model.pointnet_pack.copy_(real)
        timeit(f"mode {mode} real weights")
        z = torch.zeros_like(real)
        z[-4:] = real[-4:]          # keep the range-certificate trailer
        model.pointnet_pack.copy_(z)
        timeit(f"mode {mode} zero weights")
model.pointnet_pack.copy_(real)

# ---- the lattice-table SDF decoder (dominant kernel) -----------------------------------------------------------------
bnv.set_mlp_mode(1)
nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=1 << 21, device="cuda:0", tsdf=False)
dframes = [{"input_pts": torch.from_numpy(synthetic.frame(t)).cuda()} for t in range(40)]
for f in dframes[:30]:
    nm.integrate(f)
coords = [nm.integrate(f) for f in dframes[30:34]]
real_sdf = model.nerf.sdf_pack.clone()
CERT = 6144 + 3 * 65536 + 1024 + 256 + 1     # SD_BA + 1: certified |feature| bound of the split arithmetic


def time_decode(tag, reps=6):
    for c in coords:
        nm.volume.decode_lattice(c, model.nerf, None, query_tensor=False)
    torch.cuda.synchronize()
    lib.bnv_profile_enable(1)
    for rep in range(reps):
        for c in coords:
            nm.volume.decode_lattice(c, model.nerf, None, query_tensor=False)
    torch.cuda.synchronize()
    ms, n = (C.c_double * 4)(), (C.c_int64 * 4)()
    lib.bnv_profile_read(ms, n)
    lib.bnv_profile_enable(0)
    print(f"{tag}: lattice-table kernel {ms[1] / max(n[1], 1):.4f} ms over {n[1]} launches")


for rnd in range(2):
    model.nerf.sdf_pack.copy_(real_sdf)
    time_decode("decoder real weights")
    z = torch.zeros_like(real_sdf)
    z[CERT] = real_sdf[CERT]
    model.nerf.sdf_pack.copy_(z)
    time_decode("decoder zero weights")
model.nerf.sdf_pack.copy_(real_sdf)
for shape, sname in ((1, "16x16x32"), (0, "32x32x16")):
    for operands, name in ((1, "random f16 operands"), (0, "zero operands")):
        ms, flop = C.c_double(), C.c_double()
        lib.bnv_probe_mfma_rate(shape, operands, 16000, None, C.byref(ms), C.byref(flop))
        print(f"MFMA-only stream, {sname}, {name}: {flop.value / ms.value / 1e9:.0f} TFLOP/s ({ms.value:.2f} ms)")
