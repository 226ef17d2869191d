#!/bin/bash
# rocprofv3 kernel-trace stats of the other BASELINE configs (run on the GPU box from the repo root).
R=$PWD
export TMPDIR=/tmp
cd /tmp
for W in c3 c2 c5; do
  O=$R/gpurun_out/prof_r01_$W
  rm -rf $O; mkdir -p $O
  STEPS=8; [ $W = c5 ] && STEPS=3
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload $W --steps $STEPS --warmup 1 --cpu-seconds 0 --sampler-iterations 0 > $O/bench.json 2> $O/stats.log
  tail -c 300 $O/bench.json
done
