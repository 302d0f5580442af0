"""Do two HIP streams of one process run side by side on this box?  Chains of single-workgroup spin kernels
(~50 us each) on one stream, on two pool streams, on the default stream + a pool stream (what the frame pipeline
uses), with and without cross-stream event waits in between."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bnv_fusion_amd import _lib
lib = _lib.require_device(0)
main = torch.cuda.current_stream()
s1, s2, hp = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
CYC = 100_000     # ~50 us


def spin(st, blocks=1):
    lib.bnv_probe_spin(blocks, CYC, C.c_void_p(st.cuda_stream))


def run(sa, sb, n=40, events=False, blocks=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        spin(sa, blocks)
        spin(sb, blocks)
        if events and i % 2 == 1:
            e = torch.cuda.Event(); e.record(sa); sb.wait_event(e)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3


for name, (sa, sb) in {"one stream": (s1, s1), "two pool streams": (s1, s2), "default + pool": (main, s1),
                       "default + high-priority": (main, hp)}.items():
    run(sa, sb, 5)
    print(f"{name:26s} 80 spins of ~50 us: {run(sa, sb):6.2f} ms   with an event wait every 4: {run(sa, sb, events=True):6.2f} ms"
          f"   256 blocks each: {run(sa, sb, blocks=256):6.2f} ms")
