"""Builds libbnv_fusion_hip.so (gfx950 only) in-tree with hipcc.  No torch needed.

    python bnv_fusion_amd/csrc/build.py [--force] [--verbose]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["encode.hip", "volume.hip", "decode.hip", "frontend.hip", "tsdf.hip", "mesh.hip", "rays.hip", "io.hip", "shard.hip", "probe.hip"]
HEADERS = ["bnv_common.hpp", "frontend.hpp", os.path.join("..", "..", "include", "bnv_fusion.h")]
OUT = os.path.join(HERE, "..", "libbnv_fusion_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off",   # one rounding per float op, like the reference's ATen CPU ops
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS + ["build.py"])


def build(force=False, verbose=False):
    if not force and not needs_build():
        return os.path.abspath(OUT)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return os.path.abspath(OUT)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
