"""csrc/build.py: the library is rebuilt when -- and only when -- the SHA-256 of its sources, headers, flags and
compiler version differs from the stamp stored beside it (never by timestamp)."""
import importlib
import os
import shutil


def test_build_is_keyed_by_a_source_hash(tmp_path, monkeypatch):
    from bnv_fusion_amd.csrc import build as real
    # a scratch copy of csrc/ + include/ with a fake compiler that just writes its output file
    root = tmp_path / "pkg" / "bnv_fusion_amd" / "csrc"
    shutil.copytree(real.HERE, root)
    shutil.copytree(os.path.join(real.HERE, "..", "..", "include"), tmp_path / "pkg" / "include")
    fake = tmp_path / "hipcc"
    fake.write_text("#!/bin/sh\nif [ \"$1\" = --version ]; then echo fake-hipcc 1.0; exit 0; fi\n"
                    "while [ $# -gt 0 ]; do if [ \"$1\" = -o ]; then shift; echo lib > \"$1\"; fi; shift; done\n")
    fake.chmod(0o755)
    monkeypatch.setenv("HIPCC", str(fake))
    spec = importlib.util.spec_from_file_location("scratch_build", root / "build.py")
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for p in (b.OUT, b.STAMP):
        if os.path.exists(p):
            os.remove(p)
    assert b.needs_build()
    b.build()
    assert b.last_action == "built" and os.path.exists(b.OUT) and not b.needs_build()
    stamp = open(b.STAMP).read().strip()
    assert stamp == b.source_digest() and len(stamp) == 64
    b.build()
    assert b.last_action == "reused (hash ok)"
    # touching a file changes nothing; changing its content does
    os.utime(root / "encode.hip", None)
    b.build()
    assert b.last_action == "reused (hash ok)"
    with open(root / "bnv_common.hpp", "a") as fh:
        fh.write("// edited\n")
    assert b.needs_build()
    b.build()
    assert b.last_action == "built" and open(b.STAMP).read().strip() != stamp
    # a library without a stamp (or with a foreign one) is never trusted
    os.remove(b.STAMP)
    assert b.needs_build()
    b.build(force=False)
    assert b.last_action == "built"
    # every source listed exists in the real tree, and the real library's stamp matches the real tree
    assert all(os.path.exists(os.path.join(real.HERE, s)) for s in real.SOURCES + real.HEADERS)
    monkeypatch.delenv("HIPCC")
    if os.path.exists(real.STAMP) and os.path.exists(real._hipcc()):
        assert open(real.STAMP).read().strip() == real.source_digest()
