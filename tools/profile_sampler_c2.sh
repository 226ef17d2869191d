#!/bin/bash
R=$PWD; O=$R/gpurun_out/prof_sampler_c2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload c2 --steps 3 --warmup 1 --cpu-seconds 0 --other-workloads 0 --sampler-iterations 300 > $O/bench.json 2> $O/bench.err
f=$(ls $O/stats/*/*kernel_stats.csv | head -1); head -14 $f | cut -c1-200
python3 -c "
import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print(json.dumps(d['sampler_pipeline'])[:1200])"
