"""A long moving-camera sequence: the surrogate of BASELINE configs 0 / 2 / 4 (scene3d/lounge and ScanNet scans,
``run_e2e.py:243-252`` over ``fusion_inference_dataset.py:105-144``; the datasets are not available here).

Scene: the inside of a room, 7.2 x 2.5 x 4.7 m, with a few pieces of box furniture -- every surface an axis-aligned
box face, so a depth image is exact ray/slab arithmetic (+, -, x, /, min, max on float64: bit-reproducible on any
host, which the reference goldens of tests/golden/make_golden_sequence.py rely on).  The fusion volume (5.10 m cube,
512^3 at 1 cm; or the whole scene scaled by 1/2 for the 2.54 m / 256^3 volume) covers the middle of the room: its
x walls lie OUTSIDE the volume.  Camera: a full turn every 720 frames (0.5 degrees per frame, like the bench's pan)
with a +-12 degree nod, while it walks back and forth along the room -- about a fifth of the time it is outside the
volume, looking back into it (frames that are partly inside) or out of it (16 % of the frames have not a single point
inside: the reference's `None` path, run_e2e.py:91-92).  Over 2,000 frames the map grows to > 1 M rows from the reference's initial 100,000-row tables.

``sweep_frames`` yields the frame dicts NeuralMap takes; ``run`` drives a map over a frame source synchronously or
pipelined and returns per-frame checksums + statistics; ``bench_pass`` is bench.py's `sequence` entry (its periodic
checks against the CPU checker are the caller's: nothing here knows about it).
"""
import math
import time

import numpy as np
import torch

ROOM_HALF = (3.6, 1.25, 2.35)          # x, y (down), z half extents in metres; centred on the volume
FURNITURE = (                          # (centre, half extents): floor-standing boxes and two wall shelves
    ((-1.6, 0.90, 1.60), (0.55, 0.35, 0.45)),
    ((0.3, 0.95, -1.75), (0.80, 0.30, 0.40)),
    ((1.7, 0.80, 1.85), (0.35, 0.45, 0.30)),
    ((-0.4, 1.00, 1.95), (0.25, 0.25, 0.25)),
    ((2.1, 1.05, -1.60), (0.30, 0.20, 0.50)),
    ((-2.0, -0.45, -2.20), (0.60, 0.12, 0.15)),
    ((1.0, -0.55, 2.22), (0.70, 0.10, 0.13)),
)
YAW_STEP_DEG = 0.5
DIMS = {512: (5.10, 0.01, 1.0), 256: (2.54, 0.01, 0.5), "golden": (5.08, 0.02, 1.0)}   # volume extent, voxel, scene scale


def sweep_pose(t, scale=1.0):
    """T_wc(t): camera-to-world, camera looking along +z with y down (the datasets' convention)."""
    yaw = math.radians(180.0 + YAW_STEP_DEG * t)
    pitch = math.radians(12.0) * math.sin(2 * math.pi * t / 390.0)
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    T = np.eye(4)
    T[:3, :3] = Ry @ Rx
    T[:3, 3] = [scale * (0.9 + 2.0 * math.sin(2 * math.pi * t / 1100.0)),
                scale * 0.10 * math.sin(2 * math.pi * t / 310.0),
                scale * 0.70 * math.sin(2 * math.pi * t / 470.0)]
    return T


def intrinsics(H=480, W=640):
    from . import synthetic
    return synthetic.intrinsics(H, W)


def render_depth(T_wc, K, H, W, scale=1.0, device="cpu"):
    """Noise-free z-depth [H, W] float64 (metres) of the scene from pose T_wc: ray o + t * R (x, y, 1) meets the
    room's walls from inside (slab exit) or a furniture box from outside (slab entry); t is the z-depth."""
    dev = torch.device(device)
    f64 = dict(dtype=torch.float64, device=dev)
    u = torch.arange(W, **f64)
    v = torch.arange(H, **f64)
    x = ((u - float(K[0][2])) / float(K[0][0]))[None, :].expand(H, W)
    y = ((v - float(K[1][2])) / float(K[1][1]))[:, None].expand(H, W)
    R = np.asarray(T_wc, dtype=np.float64)[:3, :3]
    o = np.asarray(T_wc, dtype=np.float64)[:3, 3]
    d = [x * float(R[a, 0]) + y * float(R[a, 1]) + float(R[a, 2]) for a in range(3)]     # world ray direction, per axis
    inf = torch.full((H, W), float("inf"), **f64)

    def slabs(lo, hi):
        near = torch.full((H, W), -float("inf"), **f64)
        far = inf.clone()
        for a in range(3):
            inv = 1.0 / d[a]                      # +-inf for rays parallel to the slab: min / max still do the right thing
            t1 = (lo[a] - float(o[a])) * inv
            t2 = (hi[a] - float(o[a])) * inv
            near = torch.maximum(near, torch.minimum(t1, t2))
            far = torch.minimum(far, torch.maximum(t1, t2))
        return near, far

    rh = [scale * h for h in ROOM_HALF]
    _, depth = slabs([-h for h in rh], rh)                                                # inside the room: the exit
    for c, h in FURNITURE:
        lo = [scale * (c[a] - h[a]) for a in range(3)]
        hi = [scale * (c[a] + h[a]) for a in range(3)]
        near, far = slabs(lo, hi)
        hit = (near < far) & (near > 0)
        depth = torch.where(hit, torch.minimum(depth, near), depth)
    return depth


def depth_u16(t, H=480, W=640, scale=1.0, seed=0, device="cpu"):
    """The frame the depth camera stores: clean depth + N(0, 2 mm) (numpy PCG64, seeded per frame), in uint16
    millimetres like the datasets (common.py:93); 0 where nothing is closer than 60 m."""
    clean = render_depth(sweep_pose(t, scale), intrinsics(H, W), H, W, scale, device)
    noise = np.random.default_rng([seed, t]).standard_normal((H, W)) * 0.002
    mm = torch.round((clean + torch.from_numpy(noise).to(clean.device)) * 1000.0)
    mm = torch.where((mm > 0) & (mm < 60000), mm, torch.zeros_like(mm))
    return mm.to(torch.int32).to(torch.uint16)


def sweep_frames(indices, H=480, W=640, scale=1.0, device="cuda:0", render_device=None):
    """Frame dicts (``depth`` uint16 on ``device``, ``intr_mat``, ``T_wc``) of the sweep at the given time steps."""
    K = intrinsics(H, W)
    for t in indices:
        d = depth_u16(t, H, W, scale, device=render_device or device)
        yield {"frame_id": int(t), "depth": d.to(device), "intr_mat": K, "T_wc": sweep_pose(t, scale)}


def checksum(t):
    """Order-sensitive 64-bit checksum of a tensor's bit pattern (on its device; wraps around)."""
    if t is None:
        return 0
    x = t.contiguous().view(-1)
    x = x.view(torch.int32) if x.dtype == torch.float32 else x
    x = x.to(torch.int64)
    return int((x * (torch.arange(x.numel(), device=x.device, dtype=torch.int64) % 1000003 + 1)).sum().item())


def run(nm, frames, pipelined=True, in_flight=2, on_frame=None, checksums=True, check_every=None, on_check=None):
    """``NeuralMap`` over a frame source: per-frame fuse + decode (run_e2e.py:243-252's loop with the per-frame
    decode of the metric), synchronously or with ``in_flight`` frames enqueued ahead of the oldest uncollected one.
    -> dict(frames, empty_frames, rows [per frame: the row count read back behind it; exact in the pipelined
    form], capacity [initial, then after every enqueue], sums [per frame: (coords, sdf) checksums], seconds).  ``on_frame(k, frame, coords,
    sdf)`` is called for every collected frame; every ``check_every`` frames the pipeline is drained and
    ``on_check(k, frame, coords, sdf)`` sees that frame with the volume in exactly the state it was decoded from."""
    rows, sums = [], []
    caps = [nm.volume._row_capacity]           # after every enqueue (growth happens in the enqueue)
    empty = 0
    pend = []
    k_done = 0

    def collect(item):
        nonlocal empty, k_done
        fr, h = item
        c, s = h.result() if pipelined else h
        if c is None:
            empty += 1
        sums.append((checksum(c), checksum(s)) if checksums else None)
        rows.append(nm.volume._rows_known)
        if on_frame is not None:
            on_frame(k_done, fr, c, s)
        k_done += 1
        return c, s

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k, fr in enumerate(frames):
        if pipelined:
            pend.append((fr, nm.fuse_and_decode_async(fr)))
            while len(pend) > in_flight:
                collect(pend.pop(0))
            last = None
        else:
            last = collect((fr, nm.fuse_and_decode(fr)))
        caps.append(nm.volume._row_capacity)
        if check_every and k % check_every == check_every - 1:
            while pend:
                last = collect(pend.pop(0))
            if last[0] is not None and on_check is not None:
                on_check(k, fr, *last)
    while pend:
        collect(pend.pop(0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"frames": k_done, "empty_frames": empty, "rows": rows, "capacity": caps, "sums": sums, "seconds": dt}


def bench_pass(model, dev, n_frames, check=None, grid=512, check_every=200, keep=None):
    """bench.py's `sequence` entry: ``n_frames`` of the sweep through a NeuralMap that starts at the reference's
    initial capacity (sparse_volume.py:486: 100,000 rows) and grows on demand, two frames in flight, TSDF side fusion
    on; frames are rendered on the GPU ahead of the timed loop (resident inputs, like the headline).
    ``check(nm, coords, sdf) -> (max abs err, mask decisions equal, live fraction)``: the caller's checker (bench.py:
    its CPU checker), run every ``check_every`` frames with the pipeline drained."""
    import bnv_fusion_amd as bnv
    dims, voxel, scale = DIMS[grid]
    frames = list(sweep_frames(range(n_frames), scale=scale, device=dev))
    nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=100000, device=dev, tsdf=True)
    nm.inputs_resident = True
    checks = []

    def on_check(k, fr, c, s):
        if check is not None:
            checks.append((k,) + tuple(check(nm, c, s)))

    torch.cuda.reset_peak_memory_stats(dev)
    run(nm, frames, pipelined=True, in_flight=2, checksums=False, check_every=check_every, on_check=on_check)
    # (the periodic checks run on the host inside the loop: time the loop again without them for the rate)
    nm2 = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=100000, device=dev, tsdf=True)
    nm2.inputs_resident = True
    st2 = run(nm2, frames, pipelined=True, in_flight=2, checksums=False)
    rows_end = nm2.volume.num_rows()
    caps = st2["capacity"]
    if keep is not None:
        keep.append(nm2)          # (the caller goes on with the map: bench.py meshes the whole sweep volume)
    return {"frames": st2["frames"], "value": st2["frames"] / st2["seconds"], "unit": "frames/s",
            "ms_per_frame": 1e3 * st2["seconds"] / max(st2["frames"], 1),
            "what": f"moving-camera sweep of a room (bnv_fusion_amd/sequence.py), 640x480, {grid}^3 grid / voxel "
                    f"{voxel}, two frames in flight, TSDF side fusion; the volume starts at the reference's 100,000-row "
                    "capacity and grows on demand (growth steps inside the timed loop)",
            "empty_frames": st2["empty_frames"], "rows_end": rows_end, "row_capacity_end": caps[-1] if caps else None,
            "growth_steps": int(sum(1 for a, b in zip(caps, caps[1:]) if b > a)),
            "peak_device_memory_mb": torch.cuda.max_memory_allocated(dev) / 1e6,
            "parity_checks": [{"frame": k, "sdf_max_abs_err": e, "mask_decisions_equal": m, "live_fraction": l}
                              for k, e, m, l in checks]}
