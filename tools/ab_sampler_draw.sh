#!/bin/bash
# SEPAIHRD_MH_LZ=fused|ahead in the environment selects where L z is formed (inside the fused launch / ahead, beside the evaluation).
# Where the sampler's draw kernel runs (beside the evaluation on the copy stream / behind it on the main stream), per batch size:
# ms per Adaptive-Metropolis iteration of a 300-iteration device-resident run.   tools/ab_sampler_draw.sh "c2 c3 c1"
for W in ${1:-c2 c3}; do
  for M in overlap serial auto; do
    if [ $M = auto ]; then unset SEPAIHRD_MH_DRAW; else export SEPAIHRD_MH_DRAW=$M; fi
    python3 bench.py --workload $W --steps 3 --warmup 1 --cpu-seconds 0 --other-workloads 0 --sampler-iterations 300 --sampler-long-iterations 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); sp = d['sampler_pipeline']
print('$W %-8s step %.4f ms   sampler %.4f ms/iteration (two groups: %s)  %.3g proposals/s  mismatches vs strict %s' % ('$M', d['ms_per_step'], sp['ms_per_iteration'], (sp.get('ms_per_iteration_by_groups') or {}).get('2'), sp['proposals_per_s'], sp.get('accept_trace_mismatches_vs_strict')))"
  done
done
