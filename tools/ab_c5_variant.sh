for rep in 1 2; do for L in shipped tools/libsepaihrd_f194.so; do if [ $L = shipped ]; then unset SEPAIHRD_HIP_LIB; else export SEPAIHRD_HIP_LIB=$PWD/$L; fi; python3 bench.py --workload c5 --steps 5 --warmup 2 --cpu-seconds 0 --sampler-iterations 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 $L', round(d['value']), 'ms/step', round(d['ms_per_step'],4))"; done; done
