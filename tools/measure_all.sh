#!/bin/bash
# Runs on the GPU box (via gpurun, from the repo root): every measurement DESIGN.md quotes, one log per tool under
# gpurun_out/measure/.  About three minutes on one MI355X.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'timeout -k 10 1100 bash tools/measure_all.sh'
O=gpurun_out/measure
mkdir -p $O
set -o pipefail
for W in c1 c3 c2 c5; do
  python3 bench.py --workload $W --steps 10 --warmup 2 --sampler-iterations 0 --cpu-seconds 0 > $O/bench_$W.json 2> $O/bench_$W.err \
    && python3 -c "import json,sys; d=json.load(open('$O/bench_$W.json')); print('$W', round(d['value']), 'evals/s', round(d['ms_per_step'],3), 'ms/step; other arithmetic', round(d['config']['other_arith']['evals_per_s_per_gpu']))"
done
python3 tools/sweep_batch.py 2>/dev/null | tee $O/sweep_batch.log | grep -c chains
python3 tools/bench_mh.py --chains 256 4096 --iterations 400 2>/dev/null | tee $O/bench_mh.log | cut -c1-200
python3 tools/fma_accuracy.py 2>/dev/null | tee $O/fma_accuracy.log
for A in fma strict; do python3 tools/compare_lane_split.py --arith $A --chains 1024 2>/dev/null | tr '\n' ' ' | tee $O/lane_split_$A.log; echo; done
python3 tools/run_calibration.py --chains 256 --out /tmp/measure_cal 2>/dev/null | tail -1 | tee $O/run_calibration.log | cut -c1-300
python3 tools/bench_ensemble.py 2>/dev/null | tee $O/bench_ensemble.log | tail -3
