#!/bin/bash
# rocprofv3 kernel stats of the ensemble path (run on the GPU box from the repo root)
set -e
OUT=${1:-gpurun_out/ens_prof}
mkdir -p "$OUT"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$REPO/$OUT" -o ens -- python3 "$REPO/tools/bench_ensemble.py" --samples 4096 --reps 3 > "$REPO/$OUT/bench.log" 2>&1
