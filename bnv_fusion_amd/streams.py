"""Side streams that really run beside the caller's stream.

The HIP runtime multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and two
streams that land on the same queue are served strictly in submission order: the frame pipeline's encode stream then
never overlaps the main stream's decode (measured: 0.400 instead of 0.315 ms per sharded frame at world 8 when the
pool stream torch handed out shared the default stream's queue -- which depends on how many streams RCCL and torch
created before).  ``concurrent_stream`` therefore TESTS candidates -- one spin kernel on each stream, timed with
events -- and returns the first that overlaps ``main``.  bnv_fusion_amd/__init__.py also raises the queue count
when the runtime has not been initialised yet."""
import ctypes as C

import torch

from . import _lib

_SPIN_CYCLES = 240_000         # ~115 us: long against the ~15 us a cross-stream event wait costs


def _overlaps(lib, main, cand):
    def spin(st):
        _lib.check(lib.bnv_probe_spin(1, _SPIN_CYCLES, C.c_void_p(st.cuda_stream)), "bnv_probe_spin")

    def timed(both):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        spin(main)
        if both:
            spin(cand)
            done = torch.cuda.Event()
            done.record(cand)
            main.wait_event(done)
        e1.record(main)
        e1.synchronize()
        return e0.elapsed_time(e1)

    timed(True)                                   # warm-up (first launch on a stream creates its queue)
    timed(True)
    one = min(timed(False) for _ in range(2))
    two = min(timed(True) for _ in range(2))      # side by side: one + the event wait; same queue: 2 x one
    return two < 1.6 * one, one, two


def concurrent_stream(device, main=None, priority=0, exclude=()):
    """A stream on ``device`` that is served concurrently with ``main`` (default: the current stream) and with the
    streams in ``exclude``.  Falls back to the last candidate (with ``.bnv_concurrent = False``) if none passes."""
    dev = torch.device(device)
    main = main or torch.cuda.current_stream(dev)
    lib = _lib.require_device(dev.index or 0)
    cand = None
    with torch.cuda.device(dev):
        for k in range(12):
            cand = torch.cuda.Stream(device=dev, priority=priority)
            if cand.cuda_stream == main.cuda_stream or any(cand.cuda_stream == e.cuda_stream for e in exclude):
                continue
            ok = all(_overlaps(lib, other, cand)[0] for other in (main,) + tuple(exclude))
            if ok:
                cand.bnv_concurrent = True
                return cand
    cand.bnv_concurrent = False
    return cand


_PIPE_STREAMS = {}


def pipe_streams(device, main=None, n=3):
    """The frame pipeline's side streams for ``main`` on ``device`` -- (encode, front end, blend), verified concurrent
    with ``main`` and with one another -- made ONCE per (device, main stream) and shared by every FramePipe of the
    process: a second pipe (another volume, another checkpoint) gets the streams the first one verified instead of
    drawing new candidates from torch's pool, whose mapping onto the hardware queues depends on how many streams the
    process has created by then.  Pipes that share streams only add ordering between their launches."""
    dev = torch.device(device)
    main = main or torch.cuda.current_stream(dev)
    key = (dev.index or 0, int(main.cuda_stream))
    got = _PIPE_STREAMS.setdefault(key, [])
    while len(got) < n:
        got.append(concurrent_stream(dev, main, exclude=tuple(got)))
    return tuple(got[:n])
