#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel-trace stats + PMC passes
# (counters in their own runs, as the pool requires).  Outputs under gpurun_out/prof_$TAG/.
TAG=${1:-r01}
ARGS=${2:-"--steps 20 --warmup 3 --sampler-iterations 0"}
W=${3:-""}   # e.g. "--workload c3": appended to every bench invocation
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py $ARGS $W > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS $W --cpu-seconds 0 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --sampler-iterations 0 $W > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --sampler-iterations 0 $W > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --sampler-iterations 0 $W > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_f64 -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --sampler-iterations 0 $W > $O/pmc_f64.log 2>&1
cat $O/bench.json
