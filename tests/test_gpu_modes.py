"""The arithmetic mode of the MLP kernels belongs to the MODEL and travels with every call (bnv_grid_t.mlp_mode),
not through a process global: an exact-fp32 model, a split-f16 model and a tiny-cuda-nn model driven from two host
threads on two streams at the same time each produce their single-threaded bits."""
import os
import sys
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _frames():
    from bnv_fusion_amd import synthetic
    out = []
    for t in range(3):
        p = torch.from_numpy(synthetic.frame(t, H=120, W=160))
        p[..., :3] *= 0.3                                   # fit the 64^3 test volume (as __graft_entry__.smoke)
        out.append(p.to(DEV))
    return out


def _run(bnv, model, frames, n_frames, stream, sync_every_frame):
    """n_frames of fuse + decode on a map of its own, on ``stream``: every frame's (coords, sdf) and the final rows."""
    dims, voxel = np.array([1.24] * 3), 0.02
    outs = []
    with torch.cuda.stream(stream):
        nm = bnv.NeuralMap(dims, voxel, model, device=DEV)
        for k in range(n_frames):
            if sync_every_frame:
                c, s = nm.fuse_and_decode({"input_pts": frames[k % len(frames)]})
            else:
                c, s = nm.fuse_and_decode_async({"input_pts": frames[k % len(frames)]}).result()
            outs.append((c.clone(), s.clone()))
        nm.volume.to_tensor()
        outs.append((nm.volume.active_coordinates.clone(), nm.volume.features.detach().clone()))
        stream.synchronize()
    return outs


def _same(a, b):
    return len(a) == len(b) and all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(a, b))


@pytest.mark.parametrize("sync_every_frame", [True, False])
def test_three_models_two_threads_keep_their_own_arithmetic(sync_every_frame):
    import bnv_fusion_amd as bnv
    from bnv_fusion_amd import _lib
    bnv.set_mlp_mode(1)
    frames = _frames()
    voxel = 0.02
    models = {"exact": bnv.load_pretrained(device=DEV, voxel_size=voxel).set_mlp_mode(0),
              "split": bnv.load_pretrained(device=DEV, voxel_size=voxel).set_mlp_mode(1),
              "tcnn": bnv.load_pretrained(device=DEV, voxel_size=voxel, tiny_cuda=True)}
    assert [_lib.model_mode(m) for m in models.values()] == [0, 1, 2]
    assert [_lib.model_mode(m.nerf) for m in models.values()] == [0, 1, 2]
    with pytest.raises(_lib.BnvError):
        models["tcnn"].set_mlp_mode(1)
    N = 10                                                   # weights reach min_pts: the decode mask goes live
    s0 = torch.cuda.Stream(device=DEV)
    ref = {k: _run(bnv, m, frames, N, s0, sync_every_frame) for k, m in models.items()}
    # the three arithmetics are really different ones (fp32-class twins differ in the last bits, tcnn altogether)
    live = ref["exact"][N - 1][1] != voxel
    assert float(live.float().mean()) > 0.05
    assert torch.equal(ref["exact"][N - 1][0], ref["split"][N - 1][0])                 # same voxels ...
    d = (ref["exact"][N - 1][1] - ref["split"][N - 1][1]).abs()
    assert 0.0 < float(d.max()) < 1e-6                                                 # ... fp32-class, not bitwise
    assert float((ref["exact"][N][1] - ref["tcnn"][N][1]).abs().max()) > 1e-3         # other networks altogether
    # every model repeats its own bits when it runs alone ...
    assert all(_same(ref[k], _run(bnv, m, frames, N, s0, sync_every_frame)) for k, m in models.items())
    # ... and when the three run at the same time: thread A alternates the exact-fp32 and the tiny-cuda-nn model on
    # its stream, thread B runs the split-f16 model on another; the package default stays where it was
    got, errs = {}, []
    gate = threading.Barrier(2)

    def worker(kinds, stream):
        try:
            gate.wait(timeout=60)
            for rep in range(3):
                for k in kinds:
                    got[(k, rep)] = _run(bnv, models[k], frames, N, stream, sync_every_frame)
        except Exception as e:       # noqa: BLE001 -- reported by the main thread
            errs.append(e)

    ta = threading.Thread(target=worker, args=(("exact", "tcnn"), torch.cuda.Stream(device=DEV)))
    tb = threading.Thread(target=worker, args=(("split", "split"), torch.cuda.Stream(device=DEV)))
    ta.start(); tb.start(); ta.join(300); tb.join(300)
    assert not errs, errs
    assert not ta.is_alive() and not tb.is_alive()
    for (k, rep), outs in got.items():
        assert _same(ref[k], outs), (k, rep)
    assert bnv.get_mlp_mode() == 1


def test_grid_mode_field_selects_the_kernel_not_the_process_default():
    """C ABI: a grid with mlp_mode = 1 + m runs mode m whatever bnv_set_mlp_mode says; 0 follows the default; a value
    outside [0, 4] is rejected."""
    import ctypes as C
    import bnv_fusion_amd as bnv
    from bnv_fusion_amd import _lib
    bnv.set_mlp_mode(1)
    lib = _lib.load()
    model = bnv.load_pretrained(device=DEV, voxel_size=0.02)
    frames = _frames()
    dims, voxel = np.array([1.24] * 3), 0.02
    vol = bnv.SparseVolume(8, voxel, dims, 8, device=DEV)

    def enc(mode_field, default):
        bnv.set_mlp_mode(default)
        model._mlp_mode = None if mode_field == 0 else mode_field - 1
        model.nerf.mlp_mode = model._mlp_mode
        model._grid_cache = {}
        f = model.encode_pointcloud(frames[0], vol.n_xyz, vol.min_coords, vol.max_coords, voxel, return_dense=False)
        return f[0].clone()

    try:
        exact = enc(1, 1)                 # grid says exact fp32, default says split
        assert torch.equal(exact, enc(0, 0))          # default exact, grid silent
        split = enc(2, 0)                 # grid says split, default says exact
        assert torch.equal(split, enc(0, 1))
        assert not torch.equal(exact, split)
        g = _lib.Grid.from_buffer_copy(vol._grid)
        g.mlp_mode = 5
        ws = torch.zeros(int(lib.bnv_encode_workspace_bytes(1024, g.n_xyz)), dtype=torch.uint8, device=DEV)
        pts = frames[0][0, :1024].contiguous()
        assert lib.bnv_encode_begin(_lib.ptr(pts), 1024, C.byref(g), _lib.ptr(ws), ws.numel(), 1024, None) == -1
    finally:
        bnv.set_mlp_mode(1)
        model.set_mlp_mode(None)
