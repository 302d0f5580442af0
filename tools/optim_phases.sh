#!/bin/bash
# Where k_optim_step's time goes: the library built five times (-DBNV_OPTIM_PHASES=n, development probes -- the RESULTS of
# the cut-down builds are meaningless): 0 the product library / 1 classification only / 2 + forward + loss / 3 + the three
# 256-wide backward layers / 4 everything but the gradient's atomics; the optimiser loop timed with each
# (tools/optimize_profile.py: "ray_batch_step ... synchronised ms").
set -u
cd "$(dirname "$0")/.."
SRC="encode volume decode frontend tsdf mesh rays io shard pipeline probe"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function -w"
for P in 1 2 3 4; do
  [ -f tools/libbnv_optim_phase$P.so ] || /opt/rocm/bin/hipcc $FL -DBNV_OPTIM_PHASES=$P $(for f in $SRC; do echo bnv_fusion_amd/csrc/$f.hip; done) -o tools/libbnv_optim_phase$P.so
done
for P in 0 1 2 3 4; do
  L=""; [ $P -gt 0 ] && L=$PWD/tools/libbnv_optim_phase$P.so
  echo "== phases build $P (0 = the product library)"
  BNV_FUSION_LIB=$L python3 tools/optimize_profile.py 2>&1 | grep "ray_batch_step\|steps:"
done
