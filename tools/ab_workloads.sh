#!/bin/bash
# A/B of two builds of the kernel library on the saturating workloads in ONE box: tools/ab_workloads.sh <other .so> "<workloads>" [steps]
OTHER=$1; WL=${2:-"c3 c2 c5"}; STEPS=${3:-10}
for W in $WL; do
  for rep in 1 2; do
    for which in new old; do
      if [ $which = old ]; then export SEPAIHRD_HIP_LIB=$OTHER; else unset SEPAIHRD_HIP_LIB; fi
      python3 bench.py --workload $W --steps $STEPS --warmup 2 --cpu-seconds 0 --sampler-iterations 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W $which', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'strict', round(d['config']['other_arith']['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4))"
    done
  done
done
