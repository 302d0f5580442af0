"""The C ABI from a host that is not Python: examples/capi_host.cpp (hipMalloc, a hipStream_t, include/bnv_fusion.h;
no torch anywhere in the process) fuses and decodes the same frames as the Python mirror, and every frame's counters
and output checksums agree bit for bit.  Needs a real MI355X and hipcc: run with  -m gpu."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc():
    return os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("checkpoint", ["fp32", "tcnn"])
def test_cpp_host_equals_python_mirror(tmp_path, checkpoint):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU (no CPU fallback exists)")
    if not os.path.exists(_hipcc()):
        pytest.skip("hipcc not found")
    import bnv_fusion_amd as bnv
    from bnv_fusion_amd import export_packs, sequence, synthetic
    bnv.set_mlp_mode(1)
    exe = str(tmp_path / "capi_host")
    lib_dir = os.path.join(ROOT, "bnv_fusion_amd")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "capi_host.cpp"), "-o", exe, "-L" + lib_dir, "-l:libbnv_fusion_hip.so",
           "-Wl,-rpath," + lib_dir]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    # no torch in that program: it links the HIP runtime and the library only
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libbnv_fusion_hip.so" in ldd and "torch" not in ldd and "python" not in ldd

    H, W, n = 240, 320, 8
    dims, voxel = synthetic.GRID_DIMS[256]
    tc = checkpoint == "tcnn"
    model = bnv.load_pretrained(device=DEV, voxel_size=voxel, tiny_cuda=tc)
    nm = bnv.NeuralMap(np.array([dims] * 3), voxel, model, capacity=1 << 20, device=DEV)
    depths = [synthetic.depth_u16(t, H, W) for t in range(n)]
    K = synthetic.intrinsics(H, W)
    poses = [synthetic.pose(t) for t in range(n)]
    out_dir = export_packs.export(str(tmp_path / "in"), model, nm.volume, depths, K, poses, max_depth=nm.max_depth)

    want = []
    for t in range(n):
        c, s = nm.fuse_and_decode({"depth": torch.from_numpy(depths[t]).to(DEV), "intr_mat": K, "T_wc": poses[t]})
        want.append((0 if c is None else int(c.shape[0]), nm.volume.num_rows(),
                     sequence.checksum(c) % (1 << 64), sequence.checksum(s) % (1 << 64)))
    r = subprocess.run([exe, out_dir], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    lines = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("frame ")]
    assert len(lines) == n and r.stdout.strip().splitlines()[-1].startswith(f"done frames {n}")
    for t, ln in enumerate(lines):
        f = dict(zip(ln[2::2], ln[3::2]))
        got = (int(f["n_out"]), int(f["rows"]), int(f["ids"]), int(f["sdf"]))
        assert got == want[t], (t, got, want[t])
    assert want[-1][0] > 5000 and want[-1][3] != 0
