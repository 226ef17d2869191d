#!/bin/bash
# A/B of two builds of the kernel library in ONE box (boxes differ by ~3 %): tools/ab_lib.sh <other .so> <bench args...>
# prints evals/s, ms/step (fma), ms/step (strict), kernel ms per pass; "new" = the in-tree library.
OTHER=$1; shift
for rep in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export SEPAIHRD_HIP_LIB=$OTHER; else unset SEPAIHRD_HIP_LIB; fi
    python bench.py --cpu-seconds 0 --sampler-iterations 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', round(d['value']), round(d['ms_per_step'],4), round(d['config']['other_arith']['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
  done
done
