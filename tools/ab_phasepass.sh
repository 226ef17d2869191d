#!/bin/bash
# A/B of the assembly phase pass (tools/phase_pass.py; variant library built by hand) against the shipped build
V=tools/libsepaihrd_phasepass.so
export SEPAIHRD_HIP_LIB=$PWD/$V
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
unset SEPAIHRD_HIP_LIB
bash tools/ab_variants.sh "shipped $V" --steps 60 --warmup 10
for rep in 1 2; do for L in shipped $V; do if [ $L = shipped ]; then unset SEPAIHRD_HIP_LIB; else export SEPAIHRD_HIP_LIB=$PWD/$L; fi; python3 bench.py --workload c5 --steps 5 --warmup 2 --cpu-seconds 0 --sampler-iterations 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 $L', round(d['value']), 'ms/step', round(d['ms_per_step'],4))"; done; done
