#!/bin/bash
# rocprofv3 kernel stats + PMC of the fp32-state arm on the configs[4] workload (run on the GPU box from the repo root)
TAG=${1:-r02_c5_f32}
R=$PWD
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
A="--workload c5 --precision f32 --cpu-seconds 0 --sampler-iterations 0"
python3 $R/bench.py $A --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $A --steps 5 --warmup 1 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $A --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $A --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- python3 $R/bench.py $A --steps 2 --warmup 1 > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_f64 -- python3 $R/bench.py $A --steps 2 --warmup 1 > $O/pmc_f64.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc_f32 -- python3 $R/bench.py $A --steps 2 --warmup 1 > $O/pmc_f32.log 2>&1
cat $O/bench.json | cut -c1-1500
