#!/bin/bash
OTHER=$1
for rep in 1 2; do
  for which in new old; do
    if [ $which = old ]; then export SEPAIHRD_HIP_LIB=$OTHER; else unset SEPAIHRD_HIP_LIB; fi
    python3 bench.py --workload c5 --precision f32 --steps 5 --warmup 2 --cpu-seconds 0 --sampler-iterations 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5-f32 $which', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4))"
  done
done
