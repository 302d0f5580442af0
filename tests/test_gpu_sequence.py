"""Long moving-camera sequences (bnv_fusion_amd/sequence.py): the surrogate of BASELINE configs 0 / 2 / 4, whose
datasets are not available here.

* a window of the sweep in which the camera walks out of the volume and back, against what the REFERENCE ITSELF
  produced on those frames (tests/golden/sweep_256.npz, make_golden_sequence.py);
* 2,000 frames written through the dataset writer, read back through ``FusionInferenceDataset`` and fused + decoded
  at 512^3 from the reference's initial table capacity: the synchronous loop (run_e2e.py:243-252) and the pipelined
  one (two frames in flight) must agree bit for bit on every frame while the tables grow.
Needs a real MI355X: run with  -m gpu."""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SDF_TOL = 1e-4
FEAT_TOL = 1e-4


def _oracle_check(nm, coords, sdf, model_is_tcnn=False, n_voxels=512, seed=0):
    """SDF lattices of ``n_voxels`` of a frame's voxels against the CPU oracle's decode of the same volume values
    (checker only: imports oracle/).  -> (max abs err, mask decisions equal, live fraction)."""
    from oracle import bnv_oracle as orc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sd = orc.load_weights(os.path.join(root, "bnv_fusion_amd", "weights", "pointnet_fp32.npz"))
    geo = None
    if model_is_tcnn:
        geo = orc.tcnn_geo_forward(orc.load_weights(os.path.join(root, "bnv_fusion_amd", "weights",
                                                                 "pointnet_tcnn.npz"))["nerf.model.params"])
    v = nm.volume
    sel = torch.randperm(len(coords), generator=torch.Generator().manual_seed(seed))[:n_voxels].to(coords.device)
    pick = coords[sel].cpu()
    off = torch.tensor([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)])
    nbr = torch.unique((pick[:, None, :] + off[None]).reshape(-1, 3), dim=0)
    fo, wo, _ = v.query(nbr.to(coords.device))
    ovol = orc.OracleSparseVolume(8, v.voxel_size, np.asarray(v.dimensions), 8)
    present = wo[:, 0].cpu() > 0
    ovol.insert(nbr[present], fo.cpu()[present], wo.cpu()[present], torch.zeros(int(present.sum()), 1))
    with torch.no_grad():
        ref = ovol.decode_pts(orc.lattice_coords(pick.numpy()), sd, None, is_coords=True, query_tensor=False,
                              geo=geo)[0, :, :, 0]
    got = sdf[sel].cpu()
    voxel = np.float32(v.voxel_size)
    return (float((got - ref).abs().max()), bool(torch.equal(got == voxel, ref == voxel)),
            float((ref != voxel).float().mean()))



def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module", params=["split_f16", "fp32_exact"])
def bnv(request):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (no CPU fallback exists)")
    import bnv_fusion_amd
    bnv_fusion_amd.set_mlp_mode(1 if request.param == "split_f16" else 0)
    yield bnv_fusion_amd
    bnv_fusion_amd.set_mlp_mode(1)


def test_sweep_window_vs_reference_golden(bnv):
    """48 frames (every 5th of t = 430 .. 665) of the room sweep at 256^3 / voxel 0.02 / 640x480: the camera leaves
    the volume -- fewer and fewer points inside, points inside but no voxel with min_pts pairs, 17 frames without a
    point inside (`None`) -- and comes back.  Driven twice: from ``input_pts`` built by the reference's float64 host
    front end (exactly what the reference saw: SHA-256 checked) through encode_pointcloud + _integrate, and from the
    uint16 depth images through the pipelined NeuralMap (GPU front end).  Every frame's voxel ids / counts bit-exact
    against the reference, volume keys in its insertion order, weights bit-exact, features and SDF within 1e-4."""
    from bnv_fusion_amd import sequence, synthetic
    z = np.load(os.path.join(GOLDEN, "sweep_256.npz"))
    voxel, dims = float(z["voxel_size"]), z["dims"]
    H, W = [int(v) for v in z["hw"]]
    scale = sequence.DIMS["golden"][2]
    frames_t = [int(t) for t in z["frames"]]
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel)
    vol = bnv.SparseVolume(8, voxel, dims, 8, device=DEV)
    assert vol.n_xyz.tolist() == [256, 256, 256]
    # (a model of its own: a model's encode workspace serves one stream at a time)
    nm = bnv.NeuralMap(dims, voxel, bnv.load_pretrained(device=DEV, voxel_size=voxel), device=DEV)
    K = sequence.intrinsics(H, W)
    handles, kinds = [], set()
    for k, t in enumerate(frames_t):
        d16 = sequence.depth_u16(t, H, W, scale, device="cpu").numpy()
        assert _sha(d16) == str(z["depth_sha256"][k])              # the very frames the reference saw
        T = sequence.sweep_pose(t, scale)
        pts = synthetic.depth_to_input_pts(d16.astype(np.float64) / 1000.0, K, T, max_depth=3.0).astype(np.float32)[None]
        assert _sha(pts) == str(z["input_pts_sha256"][k])
        f, c, ids, g, n = model.encode_pointcloud(torch.from_numpy(pts).to(DEV), vol.n_xyz, vol.min_coords,
                                                  vol.max_coords, vol.voxel_size, return_dense=False)
        handles.append(nm.fuse_and_decode_async({"depth": torch.from_numpy(d16).to(DEV), "intr_mat": K, "T_wc": T}))
        if len(handles) > 2:
            handles[-3].result()
        if float(z["n_avg_pts"][k]) < 0:                           # the reference returned None (run_e2e.py:91-92)
            assert f is None and int(z["n_out"][k]) == 0
            kinds.add("none")
            continue
        ids_h, c_h = ids.cpu().numpy().astype(np.int64), c.cpu().numpy().reshape(-1).astype(np.int64)
        assert len(ids_h) == int(z["n_out"][k])
        if len(ids_h):
            assert np.array_equal(ids_h, np.cumsum(z[f"flat_ids_delta_{k}"].astype(np.int64))), k
        assert _sha(ids_h) + _sha(c_h) == str(z["ids_counts_sha256"][k]), k
        assert float(n) == float(z["n_avg_pts"][k])
        if f"pcounts_{k}" in z.files:
            assert np.array_equal(c_h, z[f"pcounts_{k}"].astype(np.int64))
            if len(ids_h):
                assert np.abs(f.cpu().numpy()[::16] - z[f"feats8_{k}"]).max() <= FEAT_TOL, k
        kinds.add("voxels" if len(ids_h) else "empty_output")
        vol.track_n_pts(n)
        model._integrate(vol, g, f, c)
    assert kinds == {"none", "empty_output", "voxels"}
    outs = [h.result() for h in handles]
    for k, (c, s) in enumerate(outs):                              # the pipelined depth path: same frames come out empty
        assert (c is None) == (float(z["n_avg_pts"][k]) < 0), k
        if c is not None:
            assert len(c) == int(z["n_out"][k]), k
    for v in (vol, nm.volume):
        v.to_tensor()
        assert np.array_equal(v.active_coordinates.cpu().numpy(), z["volume_keys"].astype(np.int64))   # insertion order
        assert np.array_equal(v.weights.cpu().numpy().reshape(-1), z["volume_weights"])                # bit-exact
        assert np.abs(v.features.cpu().numpy()[::16] - z["volume_feats8"]).max() <= FEAT_TOL
    assert torch.equal(vol.features, nm.volume.features)           # GPU front end == host front end, bit for bit
    origins = torch.from_numpy(z["decode_origins"].astype(np.int64)).to(DEV)
    ref = z["decode_sdf"]
    for v in (vol, nm.volume):
        got = v.decode_lattice(origins, model.nerf, None, query_tensor=False).cpu().numpy()
        assert np.array_equal(got == np.float32(voxel), ref == np.float32(voxel))
        assert np.abs(got - ref).max() <= SDF_TOL
    assert (ref != np.float32(voxel)).mean() > 0.2
    # the last frame's own pipelined decode holds the same lattices for the voxels both decoded
    c_last, s_last = outs[-1]
    key = lambda a: (a[:, 0] * 256 + a[:, 1]) * 256 + a[:, 2]
    pos = {int(v): i for i, v in enumerate(key(c_last.cpu().numpy()))}
    rows = [pos[int(v)] for v in key(z["decode_origins"].astype(np.int64))]
    assert np.abs(s_last.cpu().numpy()[rows] - ref).max() <= SDF_TOL


def test_two_thousand_frame_sweep_sync_equals_pipelined(tmp_path):
    """BASELINE config 2's shape without its dataset: a 2,000-frame 640x480 sequence written by the dataset writer
    (PNG depth + pose files), read back by FusionInferenceDataset and run through the reference's loop -- fuse + decode
    per frame -- at 512^3 / voxel 0.01 from the reference's initial 100,000-row tables, TSDF side fusion on.  The
    synchronous loop and the pipelined one (two frames in flight) see the same frames: every frame's outputs equal
    bit for bit (checksums), rows monotone, >= 1 M rows through >= 3 growth steps, most with frames in flight,
    no sticky error, bounded memory, oracle checks along the way."""
    import bnv_fusion_amd as bnv
    from bnv_fusion_amd import datasets, sequence
    bnv.set_mlp_mode(1)
    N = 2000
    dims_m, voxel, scale = sequence.DIMS[512]
    K = sequence.intrinsics()

    def depths():
        for t in range(N):
            yield sequence.depth_u16(t, scale=scale, device=DEV).cpu().numpy()

    datasets.write_sequence(str(tmp_path), "sweep/room", depths(), K, (sequence.sweep_pose(t, scale) for t in range(N)),
                            [dims_m] * 3, filter_type=0, level=1)
    data = datasets.FusionInferenceDataset(str(tmp_path), "sweep/room", device=DEV)
    assert len(data) == N and list(data.dimensions) == [dims_m] * 3
    # (one model per map: a model's encode workspace serves one stream at a time)
    maps = [bnv.NeuralMap(data.dimensions, voxel, bnv.load_pretrained(device=DEV, voxel_size=voxel), capacity=100000,
                          device=DEV, tsdf=True, max_depth=data.max_depth) for _ in range(2)]
    sync, pipe = maps
    assert sync.volume.n_xyz.tolist() == [512, 512, 512]
    checks = []
    torch.cuda.reset_peak_memory_stats(DEV)
    sums_sync, sums_pipe, rows, pend, empty = [], [], [], [], 0
    caps = [(pipe.volume._row_capacity, 0)]                  # (capacity, frames in flight) after every enqueue

    def collect(k, frame, h):
        c, s = h.result()
        sums_pipe.append((sequence.checksum(c), sequence.checksum(s)))
        rows.append(pipe.volume._rows_known)
        return c, s

    for k, frame in enumerate(data):                         # ONE pass over the files feeds both loops
        c, s = sync.fuse_and_decode(frame)                   # run_e2e.py:243-252, synchronous
        empty += c is None
        sums_sync.append((sequence.checksum(c), sequence.checksum(s)))
        pend.append((k, frame, pipe.fuse_and_decode_async(frame)))
        caps.append((pipe.volume._row_capacity, len(pend) - 1))
        while len(pend) > 2:
            collect(*pend.pop(0))
        if k % 250 == 249:                                   # drain: the volume is in the state frame k was decoded from
            while pend:
                c, s = collect(*pend.pop(0))
            if c is not None:
                checks.append((k,) + _oracle_check(pipe, c, s, n_voxels=512))
    while pend:
        collect(*pend.pop(0))
    assert sums_sync == sums_pipe                            # every frame, coords and SDF lattices, bit for bit
    n_sync, n_pipe = sync.volume.num_rows(), pipe.volume.num_rows()        # (raises on a sticky device error)
    assert n_sync == n_pipe >= 1_000_000
    assert all(b >= a for a, b in zip(rows, rows[1:])) and rows[-1] == n_pipe
    grew = [(b[0], b[1]) for a, b in zip(caps, caps[1:]) if b[0] > a[0]]
    # the tables grow for the host-side BOUND (rows known + the worst case of every frame in flight: 307,201 rows
    # each), so the reference's 100,000 rows become 0.9 M at the first frame; re-allocation + re-hash happen with
    # frames in flight
    assert len(grew) >= 3 and sum(1 for _, inflight in grew if inflight >= 1) >= 2, grew
    assert 0.05 * N < empty < 0.4 * N                                      # the camera does leave the volume
    assert torch.equal(sync.volume._row_coords[:n_sync], pipe.volume._row_coords[:n_sync])
    assert torch.equal(sync.volume._features[:n_sync], pipe.volume._features[:n_sync])
    assert torch.equal(sync.tsdf_vol.tsdf, pipe.tsdf_vol.tsdf)
    assert len(checks) >= 4
    for k, err, mask_equal, live in checks:
        assert err <= SDF_TOL and mask_equal, (k, err)
    assert max(l for _, _, _, l in checks) > 0.2                           # the decode is live
    assert torch.cuda.max_memory_allocated(DEV) < 24e9                     # two 512^3 maps + their workspaces


def test_run_e2e_example_sweep_pipelined(tmp_path, monkeypatch, capsys):
    """examples/run_e2e.py --sweep: the room sweep written in the reference's on-disk layout, read back, fused and
    decoded per frame with two frames in flight, meshed and saved (in-process: no exec from a GPU-initialised process)."""
    import importlib.util
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("run_e2e_example", os.path.join(root, "examples", "run_e2e.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "out"
    monkeypatch.setattr(sys, "argv", ["run_e2e.py", "--sweep", "60", "--grid", "256", "--decode-frames", "--pipelined",
                                      "--no-optimize", "--out", str(out)])
    try:
        mod.main()
    finally:
        import bnv_fusion_amd
        bnv_fusion_amd.set_mlp_mode(1)
    printed = capsys.readouterr().out
    assert "fused + decoded 60 frames" in printed and "speed on local fusion" in printed
    for f in ("before_optim.ply", "final.ply", "final_sparse_volume.pth", "room.npy"):
        assert (out / f).exists(), f


@pytest.mark.parametrize("argv", [
    ["--sweep", "24", "--grid", "256", "--decode-frames", "--pipelined"],                 # per-frame decode loop, then optimise
    ["--synthetic", "12", "--voxel-size", "0.02", "--height", "240", "--width", "320"],   # run_e2e.py's own loop, optimising as it goes
])
def test_run_e2e_example_with_the_global_optimiser(tmp_path, monkeypatch, capsys, argv):
    """examples/run_e2e.py WITH its optimisation steps (the README's command passes --no-optimize): both frame loops reach
    `NeuralMap.optimize`, the final mesh and `save` (the sweep loop used to lose the dataset's max_depth on the way)."""
    import importlib.util
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("run_e2e_example", os.path.join(root, "examples", "run_e2e.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "out"
    monkeypatch.setattr(sys, "argv", ["run_e2e.py"] + argv + ["--out", str(out)])
    try:
        mod.main()
    finally:
        import bnv_fusion_amd
        bnv_fusion_amd.set_mlp_mode(1)
    printed = capsys.readouterr().out
    assert "speed on global fusion" in printed
    for f in ("before_optim.ply", "final.ply", "final_sparse_volume.pth"):
        assert (out / f).exists(), f
