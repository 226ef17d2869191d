#!/bin/bash
# several builds of the kernel library in ONE box, alternating: tools/ab_variants.sh "<lib or 'shipped'> ..." [bench args]
LIBS=$1; shift
for rep in 1 2; do
  for L in $LIBS; do
    if [ $L = shipped ]; then unset SEPAIHRD_HIP_LIB; else export SEPAIHRD_HIP_LIB=$PWD/$L; fi
    python bench.py --cpu-seconds 0 --sampler-iterations 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-32s fma %.4f ms  strict %.4f ms  kernel %.4f' % ('$L', d['ms_per_step'], d['config']['other_arith']['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
