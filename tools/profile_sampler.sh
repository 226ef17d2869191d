#!/bin/bash
# rocprofv3 kernel statistics of a device-resident Adaptive-Metropolis run (which sampler kernel costs what beside the evaluation).
#   tools/profile_sampler.sh c1 2000        -> gpurun_out/prof_sampler_c1/      (BENCH_EXTRA="--chains 16384" for another batch size)
W=${1:-c1}; IT=${2:-2000}
R=$PWD; O=$R/gpurun_out/prof_sampler_$W; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload $W ${BENCH_EXTRA:-} --steps 3 --warmup 1 --cpu-seconds 0 --other-workloads 0 --sampler-iterations $IT --sampler-long-iterations 0 > $O/bench.json 2> $O/bench.err
f=$(ls $O/stats/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-160
