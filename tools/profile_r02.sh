#!/bin/bash
# Round-2 profile set (run on the GPU box from the repo root; ~5 min): rocprofv3 kernel stats + PMC passes of the
# headline (c1, fma), kernel stats of c1 strict and of c2 / c3 / c5, kernel stats + PMC of the fp32 arm on c5.
R=$PWD
export TMPDIR=/tmp
bash tools/profile_round.sh r02_c1_fma "--steps 30 --warmup 3 --sampler-iterations 0" > gpurun_out/profile_r02_c1.log 2>&1
cd /tmp
for W in "c1 --arith strict" c3 c2 c5; do
  TAG=$(echo $W | tr -d ' -' )
  O=$R/gpurun_out/prof_r02_$TAG
  rm -rf $O; mkdir -p $O
  STEPS=10; [ "$W" = c5 ] && STEPS=3
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload $W --steps $STEPS --warmup 1 --cpu-seconds 0 --sampler-iterations 0 > $O/bench.json 2> $O/stats.log
  tail -c 200 $O/bench.json; echo
done
cd $R
bash tools/profile_f32.sh r02_c5_f32 > gpurun_out/profile_r02_f32.log 2>&1
echo done
