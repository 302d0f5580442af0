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
    assert int(lib.bnv_pointnet_pack_floats()) == 34952 + 71680 // 2
    assert int(lib.bnv_sdfmlp_pack_floats()) == 6144 + 3 * 65536 + 1024 + 256 + 4 + 409600 // 2
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
