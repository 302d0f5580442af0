"""Host-side enqueue cost per frame of NeuralMap.fuse_and_decode_async in the bench's own loop (two frames in flight)
against the frame time, fp32 and tcnn networks: is the pipelined loop host-bound with the small (tcnn) networks?"""
import sys, time, numpy as np, torch
from collections import deque
sys.path.insert(0, '.')
import bnv_fusion_amd as bnv
bnv.configure_runtime()      # 8 hardware queues, before the first HIP call (streams.py)
from bnv_fusion_amd import synthetic
dims, voxel = synthetic.GRID_DIMS[256]
frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t)).cuda(), "intr_mat": synthetic.intrinsics(), "T_wc": synthetic.pose(t)} for t in range(94)]
for tc in (False, True):
    model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel, tiny_cuda=tc)
    nm = bnv.NeuralMap(np.array([dims]*3), voxel, model, capacity=1<<22, device="cuda:0", tsdf=True)
    nm.inputs_resident = True
    for t in range(30): nm.integrate(frames[t])
    def loop(n):
        enq = res = 0.0
        pend = deque()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            a = time.perf_counter(); pend.append(nm.fuse_and_decode_async(frames[30 + i % 64])); b = time.perf_counter(); enq += b - a
            if len(pend) > 2:
                a = time.perf_counter(); pend.popleft().result(); res += time.perf_counter() - a
        while pend: pend.popleft().result()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        return 1e3 * dt / n, 1e3 * enq / n, 1e3 * res / n
    loop(50)
    ms, enq, res = loop(400)
    print(f"{'tcnn' if tc else 'fp32'}: frame {ms:.3f} ms; host inside fuse_and_decode_async {enq:.3f} ms/frame, waiting in result() {res:.3f} ms/frame")
    del nm
bnv.set_mlp_mode(1)
