"""Phase breakdown of the lattice-table decode kernel (development tool).
Build the instrumented library first:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -DBNV_PHASE_PROF \
        bnv_fusion_amd/csrc/{encode,volume,decode,frontend,tsdf}.hip -o tools/libbnv_phase_prof.so
then:  BNV_FUSION_LIB=tools/libbnv_phase_prof.so python tools/phase_prof.py
Thread 0 of every workgroup accumulates shader-clock deltas between phase marks."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
os.environ.setdefault("BNV_FUSION_LIB", os.path.abspath("tools/libbnv_phase_prof.so"))
import bnv_fusion_amd as bnv  # noqa: E402
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic, _lib  # noqa: E402

NAMES = {0: "tile top (requests)", 18: "-", 1: "L0 mfma", 2: "barrier", 3: "L0 store + stage next inputs",
         4: "barrier", 5: "L1 mfma", 6: "barrier", 7: "L1 store", 8: "barrier", 9: "L2 mfma", 10: "barrier",
         11: "L2 store", 12: "barrier", 13: "L3 mfma", 14: "fc_alpha partials", 15: "barrier",
         16: "-", 17: "reduce + table write", 19: "-"}
ORDER = [0, 18, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 19]

dev = "cuda:0"
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device=dev, voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=1 << 20, device=dev, tsdf=True)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).to(dev), "intr_mat": synthetic.intrinsics(),
           "T_wc": synthetic.pose(t)} for t in range(40)]
for f in frames[:30]:
    nm.integrate(f)
lib = _lib.load()
lib.bnv_dev_phase_read.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 256)()
for f in frames[30:34]:
    nm.fuse_and_decode(f)
lib.bnv_dev_phase_read(buf)                      # reset after warm-up
n = 6
evals = 0
for f in frames[34:34 + n]:
    nm.fuse_and_decode(f)
    evals += int(nm.volume.last_lattice_evals())
lib.bnv_dev_phase_read(buf)
v = np.array(list(buf), dtype=np.float64).reshape(8, 32)
blocks = v[0, 31]
tiles = evals / 128.0
tot = v[0, :31].sum()
print(f"{n} launches, {int(blocks)} workgroups, {evals} evaluations = {tiles:.0f} tiles; "
      f"{tot / tiles:.0f} cycles per tile; columns = waves 0..7 (cycles per tile)")
for i in ORDER:
    print(f"  {NAMES[i]:28s}" + "".join(f"{v[w, i] / tiles:8.0f}" for w in range(8)) + f"   {100 * v[0, i] / tot:5.1f} % (w0)")
mf = v[:, [1, 5, 9, 13]].sum(1)
print("  MFMA phases total %          " + "".join(f"{100 * mf[w] / tot:8.1f}" for w in range(8)))
print("  MFMA pipe time of a tile = 1200 v_mfma_f32_16x16x32_f16 x 2 waves/SIMD x 16 cyc = 38400 cyc")

# ---- point encoder (k_pointnet_scatter_x): per-wave phase cycles per 32-pair tile ----
lib.bnv_dev_enc_phase_read.argtypes = [C.POINTER(C.c_ulonglong)]
eb = (C.c_ulonglong * 128)()
lib.bnv_dev_enc_phase_read(eb)
for f in frames[34:34 + n]:
    nm.integrate(f)
lib.bnv_dev_enc_phase_read(eb)
e = np.array(list(eb), dtype=np.float64).reshape(8, 16)
n_tiles = n * (307200 // 32) * 8 / 8.0          # tiles per wave index (8 waves share them evenly)
EN = ["voxelise/stage", "L1 6->128", "split 1", "L2 128->128", "split 2", "L3 128->128", "split 3", "L4 128->8", "scatter"]
etot = e[:, :9].sum(1)
print(f"encoder: {etot.mean() / n_tiles:.0f} cycles per tile per wave; MFMA pipe time = 456 v_mfma_f32_16x16x32_f16 x 16 cyc = 7300 per wave, two waves per SIMD")
for i, name in enumerate(EN):
    print(f"  {name:16s}" + "".join(f"{e[w, i] / n_tiles:8.0f}" for w in range(8)) + f"   {100 * e[:, i].sum() / etot.sum():5.1f} %")
