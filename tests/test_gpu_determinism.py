"""Repeatability under load: every stage of the frame path gives the same bits when it is run again on the same state,
at the full frame size (640x480, 256^3 / 512^3) where every workgroup-level hand-over in the kernels (LDS entry
buffers and their flushes, look-back scans, dynamic tile draws) is exercised tens of thousands of times per launch.
A wave that falls out of step with its workgroup shows up here as a differing run, not as a wrong golden: the small
golden cases rarely hit such a window.  Needs a real MI355X: run with  -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", params=["split_f16", "tcnn"])
def setup(request):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (no CPU fallback exists)")
    import bnv_fusion_amd as bnv
    from bnv_fusion_amd import sequence
    bnv.set_mlp_mode(1)
    dims, voxel, scale = sequence.DIMS[512]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel, tiny_cuda=request.param == "tcnn")
    frames = list(sequence.sweep_frames(range(0, 36), scale=scale, device=DEV))
    yield bnv, model, frames, dims, voxel
    bnv.set_mlp_mode(1)


def _map(bnv, model, dims, voxel, capacity=1 << 21):
    return bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=capacity, device=DEV)


def test_lattice_decode_repeats_bit_for_bit(setup):
    """decode_lattice of one frame's voxels, 60 times on an unchanged volume, with the neighbour pass separate and
    fused into the mark kernel: all runs equal (the frame holds ~100 k voxels: ~2,600 chunks of the mark kernel with
    an LDS flush each)."""
    from bnv_fusion_amd import _lib
    bnv, model, frames, dims, voxel = setup
    nm = _map(bnv, model, dims, voxel)
    for f in frames[:24]:
        nm.integrate(f)
    lib = _lib.load()
    try:
        for f in frames[24:30]:
            c = nm.integrate(f)
            assert c is not None and c.shape[0] > 50000
            ref = None
            for fused in (0, 1):
                assert lib.bnv_set_option(b"fused_mark", fused) == 0
                for rep in range(60):
                    out = nm.volume.decode_lattice(c, model.nerf, None, query_tensor=False)
                    if ref is None:
                        ref = out.clone()
                    assert torch.equal(out, ref), (f["frame_id"], fused, rep, int((out != ref).any(1).sum()))
    finally:
        lib.bnv_set_option(b"fused_mark", -1)


def test_encode_and_integrate_repeat_bit_for_bit(setup):
    """The encoder (voxel order, counts, features) 10 times per frame, and the same 12 frames upserted into three
    fresh volumes: keys in the same insertion order, weights and features equal."""
    bnv, model, frames, dims, voxel = setup
    nm = _map(bnv, model, dims, voxel)
    v = nm.volume
    for f in frames[30:34]:
        ref = None
        for rep in range(10):
            feats, pcounts, flat_ids, grid_ids, counters, cap, _ = model.encode_depth_async(
                f["depth"], f["intr_mat"], f["T_wc"], nm.max_depth, v.n_xyz, v.min_coords, v.max_coords, v.voxel_size)
            n_out = int(counters[2].item())
            assert n_out > 50000
            out = [feats[:n_out].clone(), pcounts[:n_out].clone(), grid_ids[:n_out].clone(), counters.clone()]
            if ref is None:
                ref = out
            for a, b in zip(out, ref):
                assert torch.equal(a, b), (f["frame_id"], rep)
    states = []
    for trial in range(3):
        m = _map(bnv, model, dims, voxel, capacity=100000)      # grows on the way
        for f in frames[:12]:
            m.integrate(f)
        states.append([t.clone() for t in m.volume.to_tensor()])
    assert states[0][0].shape[0] > 100000
    for s in states[1:]:
        for a, b in zip(s, states[0]):
            assert torch.equal(a, b)


def test_pipelined_frames_repeat_bit_for_bit(setup):
    """The whole pipelined frame (two streams, three frames in flight) over 30 frames, three times from an empty
    map: per-frame checksums of the voxel lists and SDF lattices equal."""
    from bnv_fusion_amd import sequence
    bnv, model, frames, dims, voxel = setup
    runs = []
    for trial in range(3):
        nm = _map(bnv, model, dims, voxel, capacity=100000)
        nm.inputs_resident = True
        st = sequence.run(nm, frames[:30], pipelined=True, in_flight=3)
        runs.append(st["sums"])
    assert runs[1] == runs[0] and runs[2] == runs[0]


def test_finalize_strides_over_its_tiles(setup):
    """The encoder's compaction kernel on 5 workgroups (every workgroup strides over ~100 tiles of the look-back
    chain) and on its default grid: the same voxel order, counts and features."""
    from bnv_fusion_amd import _lib
    bnv, model, frames, dims, voxel = setup
    nm = _map(bnv, model, dims, voxel)
    v = nm.volume
    lib = _lib.load()
    try:
        for f in frames[20:24]:
            outs = []
            for blocks in (0, 5, 1, 0):
                assert lib.bnv_set_option(b"finalize_blocks", blocks) == 0
                feats, pcounts, flat_ids, grid_ids, counters, cap, _ = model.encode_depth_async(
                    f["depth"], f["intr_mat"], f["T_wc"], nm.max_depth, v.n_xyz, v.min_coords, v.max_coords, v.voxel_size)
                n_out = int(counters[2].item())
                assert n_out > 50000
                outs.append([feats[:n_out].clone(), pcounts[:n_out].clone(), grid_ids[:n_out].clone(), counters.clone()])
            for o in outs[1:]:
                for a, b in zip(o, outs[0]):
                    assert torch.equal(a, b), f["frame_id"]
    finally:
        lib.bnv_set_option(b"finalize_blocks", 0)
