"""Do CU-masked HIP streams (hipExtStreamCreateWithCUMask, through bnv_stream_create_cu_mask) partition this GPU?

1. one masked stream at a time: 4,096 single-wave spin workgroups (16 per CU on the whole device) that each spin for
   ~50 us -- the elapsed time tells how many CUs served them;
2. two streams with disjoint masks at once: do they run side by side without taking each other's CUs?
3. the MFMA probe kernel (one 512-thread workgroup per CU of the DEVICE) on masked streams: time ~ 256 / CUs.

Diagnostic only (GPU box): python3 tools/cu_mask_probe.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bnv_fusion_amd as bnv  # noqa: E402
from bnv_fusion_amd import _lib  # noqa: E402

bnv.configure_runtime()
dev = torch.device("cuda:0")
lib = _lib.require_device(0)
CUS = torch.cuda.get_device_properties(0).multi_processor_count
WORDS = (CUS + 31) // 32


def mask_of(bits):
    m = (C.c_uint32 * WORDS)()
    for b in bits:
        m[b // 32] |= 1 << (b % 32)
    return m


def masked_stream(bits):
    out = C.c_void_p()
    _lib.check(lib.bnv_stream_create_cu_mask(WORDS, mask_of(bits), C.byref(out)), "bnv_stream_create_cu_mask")
    return torch.cuda.ExternalStream(out.value, device=dev)


def spin(st, blocks, cycles=100_000):
    _lib.check(lib.bnv_probe_spin(blocks, cycles, C.c_void_p(st.cuda_stream)), "bnv_probe_spin")


def timed(streams, blocks):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in streams]
    torch.cuda.synchronize()
    for (a, b), st, n in zip(ev, streams, blocks):
        a.record(st)
        spin(st, n)
        b.record(st)
    torch.cuda.synchronize()
    first = ev[0][0]
    return [a.elapsed_time(b) for a, b in ev], max(first.elapsed_time(b) for _, b in ev)


print(f"{CUS} CUs, {WORDS} mask words")
plain = torch.cuda.Stream(device=dev)
for _ in range(2):
    t, _w = timed([plain], [4096])
print(f"unmasked stream, 4096 spin workgroups: {t[0]:.3f} ms")
layouts = {
    "low bits": lambda n: range(n),
    "high bits": lambda n: range(CUS - n, CUS),
    "every (256/n)-th": lambda n: range(0, CUS, CUS // n),
}
for name, f in layouts.items():
    for n in (256, 192, 128, 64, 32):
        if n > CUS:
            continue
        st = masked_stream(list(f(n)))
        timed([st], [4096])
        t, _w = timed([st], [4096])
        print(f"mask {name:18s} {n:4d} CUs: {t[0]:.3f} ms  (x{t[0] / 1.0:.2f})")
# two disjoint masks at once
for na in (128, 192, 64):
    a = masked_stream(list(range(na)))
    b = masked_stream(list(range(na, CUS)))
    timed([a, b], [16 * na, 16 * (CUS - na)])
    t, w = timed([a, b], [16 * na, 16 * (CUS - na)])
    ta, _ = timed([a], [16 * na])
    tb, _ = timed([b], [16 * (CUS - na)])
    print(f"disjoint {na} + {CUS - na} CUs, 16 spin workgroups per CU each: together {t[0]:.3f} / {t[1]:.3f} ms "
          f"(wall {w:.3f}), alone {ta[0]:.3f} / {tb[0]:.3f} ms")
# overlapping: masked + unmasked
a = masked_stream(list(range(64)))
t, w = timed([a, plain], [1024, 4096])
print(f"masked 64 (1024 wgs) + unmasked (4096 wgs) together: {t[0]:.3f} / {t[1]:.3f} ms, wall {w:.3f}")
# the MFMA probe on masked streams
for n in (256, 128, 64):
    st = masked_stream(list(range(n)))
    ms, fl = C.c_double(), C.c_double()
    for _ in range(2):
        _lib.check(lib.bnv_probe_mfma_rate(1, 1, 2000, C.c_void_p(st.cuda_stream), C.byref(ms), C.byref(fl)), "probe")
    print(f"MFMA probe ({CUS} workgroups of 512) on a {n}-CU mask: {ms.value:.3f} ms")
