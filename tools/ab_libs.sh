#!/bin/bash
# A/B of library builds: tools/ab_libs.sh lib1.so lib2.so ... (each run: bench.py, 40 steps; 3 interleaved rounds)
for r in 1 2 3; do
  for l in "$@"; do
    BNV_FUSION_LIB=$PWD/$l python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt-mode 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', 'fps %.1f' % d['value'], 'decode_ms %.4f' % d['roofline']['avg_kernel_ms'], 'enc_ms %.4f' % d['kernels']['pointnet_scatter']['avg_ms'])"
  done
done
