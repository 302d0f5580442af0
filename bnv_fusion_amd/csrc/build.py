"""Builds libbnv_fusion_hip.so (gfx950 only) in-tree with hipcc.  No torch needed.

    python bnv_fusion_amd/csrc/build.py [--force] [--verbose]

The library carries a stamp file next to it (libbnv_fusion_hip.so.sha256): the SHA-256 of every source, header, the
compiler flags and the hipcc version it was built from.  build() recompiles whenever that digest differs from the
tree's -- never by timestamp -- and says which it did (`built` / `reused (hash ok)`).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["encode.hip", "volume.hip", "decode.hip", "frontend.hip", "tsdf.hip", "mesh.hip", "rays.hip", "io.hip",
           "shard.hip", "pipeline.hip", "probe.hip"]
HEADERS = ["bnv_common.hpp", "frontend.hpp", os.path.join("..", "..", "include", "bnv_fusion.h")]
OUT = os.path.join(HERE, "..", "libbnv_fusion_hip.so")
STAMP = OUT + ".sha256"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off",   # one rounding per float op, like the reference's ATen CPU ops
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]
last_action = None              # "built" | "reused (hash ok)" after build()


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def source_digest():
    """SHA-256 over the sources, headers, flags and the compiler's version string."""
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        h.update(f.encode())
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(hashlib.sha256(fh.read()).digest())
    h.update(" ".join(FLAGS).encode())
    try:
        ver = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception:
        ver = "hipcc unavailable"
    h.update(ver.encode())
    return h.hexdigest()


def needs_build():
    if not (os.path.exists(OUT) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_digest()


def build(force=False, verbose=False):
    global last_action
    digest = source_digest()
    if not force and os.path.exists(OUT) and os.path.exists(STAMP) and open(STAMP).read().strip() == digest:
        last_action = "reused (hash ok)"
        print(f"libbnv_fusion_hip.so: {last_action} {digest[:12]}", file=sys.stderr)
        return os.path.abspath(OUT)
    cmd = [_hipcc()] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    if os.path.exists(STAMP):
        os.remove(STAMP)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(digest + "\n")
    last_action = "built"
    print(f"libbnv_fusion_hip.so: {last_action} {digest[:12]}", file=sys.stderr)
    return os.path.abspath(OUT)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
