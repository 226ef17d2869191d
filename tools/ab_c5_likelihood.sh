#!/bin/bash
# Where the likelihood of the saturating workloads is evaluated, A/B in one box with the experiment build
# (csrc/Makefile `experiments`): inline in the integrator against the separate pass over parked increments, and for the
# separate pass the (chain, stream)-serial walk against the (chain, day, age)-parallel kernel.  usage: tools/ab_c5_likelihood.sh [steps]
STEPS=${1:-6}
export SEPAIHRD_HIP_LIB=$PWD/tools/libsepaihrd_hip_experiments.so
run() {  # label, workload, env...
  local label=$1 w=$2; shift 2
  env "$@" python bench.py --workload $w --steps $STEPS --warmup 2 --cpu-seconds 0 --sampler-iterations 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-44s %s  %8.3f ms/step  %9.0f evals/s  integrator %8.3f ms  likelihood pass %6.3f ms  (%s)' % ('$label', '$w', d['ms_per_step'], d['value'], r['kernel_ms'], r['likelihood_pass_ms'], r['likelihood_form']))"
}
for rep in 1 2; do
  run "c5 inline"                          c5 SEPAIHRD_SPLIT_LL=0
  run "c5 separate pass, serial walk"      c5 SEPAIHRD_SPLIT_LL=1 SEPAIHRD_LL_SERIAL_MIN_WAVES=0
  run "c5 separate pass, parallel terms"   c5 SEPAIHRD_SPLIT_LL=1 SEPAIHRD_LL_SERIAL_MIN_WAVES=100000000
  run "c3 inline"                          c3 SEPAIHRD_SPLIT_LL=0
  run "c3 separate pass, serial walk"      c3 SEPAIHRD_SPLIT_LL=1 SEPAIHRD_LL_SERIAL_MIN_WAVES=0
  run "c3 separate pass, parallel terms"   c3 SEPAIHRD_SPLIT_LL=1 SEPAIHRD_LL_SERIAL_MIN_WAVES=100000000
done
unset SEPAIHRD_HIP_LIB
run "c5 shipped library" c5 X=1
run "c3 shipped library" c3 X=1
run "c2 shipped library" c2 X=1
