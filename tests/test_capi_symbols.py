"""The C-ABI library loads and exports every entry point include/bnv_fusion.h declares.
CPU only: no compute call is made."""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "bnv_fusion.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bnv_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from bnv_fusion_amd import _lib
    assert _declared() == sorted(_lib.SYMBOLS)


def test_library_exports_every_symbol():
    from bnv_fusion_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert int(lib.bnv_pointnet_pack_floats()) == 34952 + 77824 // 2 + 4      # + split pack + certified-range trailer
    assert int(lib.bnv_sdfmlp_pack_floats()) == 6144 + 3 * 65536 + 1024 + 256 + 4 + 2 * (409600 // 2)
    assert lib.bnv_status_string(0) == b"ok"
    # without bnv_init every compute entry refuses to run instead of silently doing nothing
    import ctypes as C
    g = _lib.Grid()
    rc = lib.bnv_encode_pointcloud(C.c_void_p(8), 1, C.byref(g), C.c_void_p(8), C.c_void_p(8), 1 << 30, 1,
                                   None, None, None, None, 0, 0, C.c_void_p(8), None)
    assert rc != 0


def test_product_has_no_cpu_fallback():
    """The product path must fail loudly off-GPU, not compute on the CPU."""
    import torch
    import bnv_fusion_amd as b
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = b.load_pretrained(device="cpu")
    pts = torch.zeros((1, 10, 6))
    with pytest.raises(Exception):
        m.encode_pointcloud(pts, [64, 64, 64], [-0.64] * 3, [0.64] * 3, 0.02, return_dense=False)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "bnv_fusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f


def test_argument_validation_without_a_gpu():
    """Invalid arguments are refused before any HIP call (status codes of include/bnv_fusion.h), so this runs on CPU."""
    import ctypes as C
    from bnv_fusion_amd import _lib
    lib = _lib.load()
    INVALID = -1
    v = _lib.Volume()                                   # all-null volume
    assert lib.bnv_volume_integrate(C.byref(v), None, None, None, 4, None, None, 0, None) == INVALID
    assert lib.bnv_volume_integrate_batch(C.byref(v), 1, None, None, None, None, None, None, None, None, 0, None) == INVALID
    assert lib.bnv_volume_clear(C.byref(v), None) == INVALID
    assert lib.bnv_tsdf_integrate_u16(None, None, None, None, None, 0.025, 0.125, None, None, 480, 640, None, None, 1.0,
                                      3.0, None, None) == INVALID
    assert lib.bnv_tsdf_integrate_batch_u16(C.c_void_p(8), C.c_void_p(8), None, (C.c_int32 * 3)(4, 4, 4),
                                            (C.c_float * 3)(), 0.025, 0.125, 9, C.c_void_p(8), None, 4, 4,
                                            (C.c_float * 9)(), (C.c_float * 16)(), 1.0, 3.0, None) == INVALID   # > 8 frames
    assert lib.bnv_depth_to_points(None, 0, 480, 640, None, None, 10.0, None, 0, None, None, None) == INVALID
    assert lib.bnv_png_unfilter(None, 1, 1, 2, None) != 0
    assert lib.bnv_set_option(b"no_such_option", 1) == INVALID
    assert lib.bnv_set_mlp_mode(7) == INVALID
    for code in (0, -1, -2, -3, -4, -5):
        assert len(lib.bnv_status_string(code)) > 0
    assert int(lib.bnv_volume_workspace_bytes(1000)) > 8000
    assert int(lib.bnv_depth_workspace_bytes(480, 640)) > 0


def test_missing_library_fails_loudly(tmp_path):
    """No library -> ImportError naming the build command; nothing falls back to a CPU path."""
    import subprocess
    import sys
    code = ("import bnv_fusion_amd as b, torch\n"
            "try:\n"
            "    b.load_pretrained(device='cpu').encode_pointcloud(torch.zeros((1, 4, 6)), [8, 8, 8], [0.] * 3, [1.] * 3, 0.125)\n"
            "except ImportError as e:\n"
            "    assert 'no CPU fallback' in str(e), e\n"
            "    print('LOUD')\n")
    env = dict(os.environ, BNV_FUSION_LIB=str(tmp_path / "absent.so"), PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "LOUD" in out.stdout, out.stdout + out.stderr


def test_header_is_plain_c_and_cxx():
    """include/bnv_fusion.h is what a cgo / JNI / N-API / C host includes: it must compile as C99 and as C++11 on its
    own (no HIP, no torch types in any signature)."""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, "include", "bnv_fusion.h")
    for cc, args in (("gcc", ["-x", "c", "-std=c99"]), ("g++", ["-x", "c++", "-std=c++11"])):
        if not shutil.which(cc):
            pytest.skip(f"{cc} not found")
        r = subprocess.run([cc, "-fsyntax-only", "-Wall", "-Wextra", "-Werror"] + args + [hdr],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    txt = open(hdr).read()
    assert "#include <hip" not in txt and "at::" not in txt and "c10::" not in txt


def test_cpp_host_example_builds_against_the_library(tmp_path):
    """examples/capi_host.cpp (the hot path from a host without Python or torch) compiles and links against the
    in-tree library; the GPU test runs it (tests/test_gpu_capi_host.py)."""
    import shutil
    import subprocess
    from bnv_fusion_amd import _lib
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "capi_host")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "capi_host.cpp"), "-o", exe, "-L" + lib_dir,
                        "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + lib_dir],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libbnv_fusion_hip" in ldd and "torch" not in ldd and "python" not in ldd
