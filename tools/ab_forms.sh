#!/bin/bash
# One box: bit-identity of the three kernel forms, then the latency of the 16-lane form against the one-wavefront-per-chain form.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
{
timeout -k 10 300 python tools/compare_lane_split.py --solver 0 --chains 100
timeout -k 10 300 python tools/compare_lane_split.py --solver 1 --chains 100
timeout -k 10 400 python tools/compare_lane_split.py --solver 0 --fuzz 7 --fuzz-cases 16
timeout -k 10 400 python tools/compare_lane_split.py --solver 1 --fuzz 8 --fuzz-cases 16
for w in 0 2 0 2; do
  echo "== SEPAIHRD_WAVE_CHAIN=$w"
  SEPAIHRD_WAVE_CHAIN=$w SWEEP_SIZES=1,16,63,256,1024,2048,4096 timeout -k 10 300 python tools/sweep_batch.py
done
} > gpurun_out/ab_forms.log 2>&1
