#!/bin/bash
# Diagnostic: issue / wait / instruction-fetch counters of the 16-lane evaluation kernel (both arithmetic builds) for several
# builds of the library in ONE box: tools/pmc_placement.sh "<lib or 'shipped'> ..."   (counters in their own passes)
LIBS=$1
R=$PWD
O=$R/gpurun_out/pmc_placement
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for L in $LIBS; do
  T=$(basename $L .so)
  if [ $L = shipped ]; then unset SEPAIHRD_HIP_LIB; else export SEPAIHRD_HIP_LIB=$R/$L; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/$T/p1 -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --sampler-iterations 0 > $O/$T.p1.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/$T/p2 -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --sampler-iterations 0 > $O/$T.p2.log 2>&1 || exit 1
  python3 $R/tools/pmc_placement_summary.py $T $O/$T
  find $O/$T -name "*.csv" -size +2M -delete
done
