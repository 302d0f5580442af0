"""Samples rocm-smi (sclk, power) while the fuse+decode loop runs back to back for a few seconds."""
import subprocess, sys, threading, time
import numpy as np, torch
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic
dims, voxel = synthetic.GRID_DIMS[256]
model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel)
nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<20, device="cuda:0", tsdf=True)
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(60)]
for t in range(30): nm.integrate(frames[t])
stop = False
def sample():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            l = [x.strip() for x in o.splitlines() if ("sclk" in x or "Power" in x or "power" in x)]
            print(round(time.time() - t0, 2), " | ".join(l)[:300], flush=True)
        except Exception as e:
            print("smi failed", e); return
        time.sleep(0.4)
t0 = time.time()
th = threading.Thread(target=sample); th.start()
time.sleep(1.0)
print("--- load starts", round(time.time() - t0, 2), flush=True)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(8):
    ev0.record()
    for t in range(30, 60): nm.fuse_and_decode(frames[t])
    ev1.record(); torch.cuda.synchronize()
    print(f"rep {rep}: {ev0.elapsed_time(ev1)/30:.3f} ms/frame at t={time.time()-t0:.2f}", flush=True)
stop = True; th.join()
