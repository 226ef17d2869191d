#!/bin/bash
# Round 3: rocprofv3 kernel-trace stats of the self-contained device sampler (streams, accept test, scale adaptation on the
# device) -- the headline problem at 4096 chains x 20 000 iterations and configs[2] at 65 536 chains x 1500 -- plus the
# device-side period between evaluation starts.  Run on the GPU box from the repo root; summaries go to profiles/ by hand.
R=$PWD
O=$R/gpurun_out/prof_sampler_r03
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c1 -- python3 $R/tools/long_run_sampler.py --iterations 20000 > $O/c1_run.json 2> $O/c1.log || exit 1
cut -c1-180 $O/c1/*/*kernel_stats.csv | head -14
python3 $R/tools/sampler_timeline.py $(ls $O/c1/*/*kernel_trace.csv | head -1) > $O/c1_timeline.txt 2>&1
cat $O/c1_timeline.txt | tail -15
rm -f $O/c1/*/*kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 $R/tools/long_run_sampler.py --workload c2 --chains 65536 --iterations 1500 --burn-in 300 --thinning 500 > $O/c2_run.json 2> $O/c2.log || exit 1
cut -c1-180 $O/c2/*/*kernel_stats.csv | head -14
python3 $R/tools/sampler_timeline.py $(ls $O/c2/*/*kernel_trace.csv | head -1) > $O/c2_timeline.txt 2>&1
tail -15 $O/c2_timeline.txt
rm -f $O/c2/*/*kernel_trace.csv
tail -c 600 $O/c1_run.json; tail -c 600 $O/c2_run.json
