#!/bin/bash
# rocprofv3 kernel-trace stats of the device-resident sampler loop + the device-side period between evaluation starts
# (run on the GPU box from the repo root; the summaries go to profiles/ by hand: r02_sampler_*)
R=$PWD
O=$R/gpurun_out/prof_sampler
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_mh.py --chains 4096 --iterations 400 --state device > $O/bench.json 2> $O/stats.log
cd $R
cat $O/bench.json | tail -1 | cut -c1-300
cat $O/stats/*/*kernel_stats.csv | cut -c1-200 | head -20
python3 tools/sampler_timeline.py $(ls $O/stats/*/*kernel_trace.csv | head -1) | tee $O/timeline.txt
