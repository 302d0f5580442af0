#!/bin/bash
# A/B of the whole tree against an extracted older tree (tools/_r02_tree, not committed): interleaved bench runs on one box
for r in 1 2 3; do
  for t in tools/_r02_tree .; do
    extra=""; [ "$t" = "." ] && extra="--preheat 0 --sequence-frames 0"
    python $t/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt-mode $extra "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t'.ljust(18), 'fps %.1f' % (d['burst']['value'] if 'burst' in d else d['value']), 'decode_ms %.4f' % d['roofline']['avg_kernel_ms'], 'enc_ms %.4f' % d['kernels']['pointnet_scatter']['avg_ms'])"
  done
done
