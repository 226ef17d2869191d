#!/bin/bash
# A/B of the small-batch likelihood forms in ONE box (boxes differ by ~2 %): prints evals/s, ms/step (fma), ms/step
# (strict), kernel ms, likelihood-pass ms
for rep in 1 2 3; do
  for mode in 1 0; do
    SEPAIHRD_FUSED_LL=$mode python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --sampler-iterations 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$mode', round(d['value']), round(d['ms_per_step'],4), round(d['config']['other_arith']['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['likelihood_pass_ms'],4))"
  done
done
