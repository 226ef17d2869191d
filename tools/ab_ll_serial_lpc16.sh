export SEPAIHRD_HIP_LIB=$PWD/tools/libsepaihrd_hip_experiments.so
for ch in 8192 16384; do for m in 0 100000000; do
 SEPAIHRD_SPLIT_LL=1 SEPAIHRD_LL_SERIAL_MIN_WAVES=$m python bench.py --workload c5 --chains $ch --steps 6 --warmup 2 --cpu-seconds 0 --sampler-iterations 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('chains $ch serial_min_waves $m: %8.3f ms/step integrator %8.3f likelihood pass %6.3f ms (%s)' % (d['ms_per_step'], r['kernel_ms'], r['likelihood_pass_ms'], r['likelihood_form']))"
done; done
