#!/bin/bash
# A/B of builds of the kernel library in ONE box (boxes of the pool differ by ~4 %, so only same-box alternating runs compare).
#   tools/ab.sh [-w "c1 c3 c5"] [-s steps] [-r reps] [-a "extra bench args"] <lib> [<lib> ...]
# <lib> = `shipped` (the in-tree library) or a path to a variant (make -C .../csrc variant NAME=x EXTRA="-D..." -> tools/libsepaihrd_x.so).
# One line per (workload, rep, lib): evals/s, ms per step in the run's arithmetic and in the other one, the kernel's own ms.
# Replaces the round-1..3 scripts ab_bench / ab_lib / ab_variants / ab_workloads / ab_forms / ab_f32 / ab_c5_* / ab_phase*.
WL="c1"; STEPS=20; REPS=2; ARGS=""
while getopts "w:s:r:a:" o; do
  case $o in w) WL=$OPTARG;; s) STEPS=$OPTARG;; r) REPS=$OPTARG;; a) ARGS=$OPTARG;; *) exit 2;; esac
done
shift $((OPTIND - 1))
[ $# -ge 1 ] || { echo "usage: $0 [-w workloads] [-s steps] [-r reps] [-a bench-args] lib..." >&2; exit 2; }
for W in $WL; do
  for rep in $(seq 1 $REPS); do
    for L in "$@"; do
      if [ "$L" = shipped ]; then unset SEPAIHRD_HIP_LIB; else export SEPAIHRD_HIP_LIB=$PWD/$L; fi
      python3 bench.py --workload $W --steps $STEPS --warmup 3 --cpu-seconds 0 --sampler-iterations 0 --other-workloads 0 $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-3s %-34s %10.0f evals/s  %8.4f ms/step  other arith %8.4f  kernel %8.4f + ll %6.4f  vgprs %d' % ('$W', '$L', d['value'], d['ms_per_step'], d['config']['other_arith']['ms_per_step'], r['kernel_ms'], r['likelihood_pass_ms'], d['kernel_info']['vgprs']))"
    done
  done
done
