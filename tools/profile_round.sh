#!/bin/bash
# Runs on the GPU box (via gpurun, from the repo root): for one workload the bench line, the rocprofv3 kernel-trace stats of the
# same command and the PMC passes (counters in their own runs, as the pool requires: --pmc never together with a trace domain
# other than --kernel-trace).  Outputs under gpurun_out/prof_$TAG/; tools/summarize_profile.py $TAG condenses them into profiles/.
#   tools/profile_round.sh r04_c3_fma "--workload c3 --steps 10 --warmup 2"
TAG=${1:-r04_c1_fma}
ARGS=${2:-"--steps 20 --warmup 3"}
COMMON="--cpu-seconds 0 --sampler-iterations 0 --other-workloads 0"
R=$PWD
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py $ARGS $COMMON > $O/bench.json 2> $O/bench.err || { tail -c 600 $O/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS $COMMON > $O/stats.log 2>&1
PM="--steps 3 --warmup 1 $(echo $ARGS | sed -E 's/--steps [0-9]+//; s/--warmup [0-9]+//')"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $PM $COMMON > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $PM $COMMON > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- python3 $R/bench.py $PM $COMMON > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_f64 -- python3 $R/bench.py $PM $COMMON > $O/pmc_f64.log 2>&1
python3 -c "
import json; d = json.loads([l for l in open('$O/bench.json') if l.startswith('{')][-1]); print('$TAG', round(d['value']), 'evals/s', round(d['ms_per_step'], 4), 'ms/step kernel', round(d['roofline']['kernel_ms'], 4), 'frac', round(d['roofline']['frac'], 4))"
