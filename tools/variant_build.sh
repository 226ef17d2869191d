#!/bin/bash
# Experiment builds of the kernel library: tools/variant_build.sh <name> <extra hipcc flags...>
set -e
NAME=$1; shift
cd "$(dirname "$0")/../mathematical-modeling-of-infectious-diseases-v1_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -I../../include -I. $@"
/opt/rocm/bin/hipcc $F -ffp-contract=off -DSEPAIHRD_ARITH_FMA=0 -c sepaihrd_kernels.hip -o /tmp/kv_${NAME}_strict.o
/opt/rocm/bin/hipcc $F -ffp-contract=fast -DSEPAIHRD_ARITH_FMA=1 -c sepaihrd_kernels.hip -o /tmp/kv_${NAME}_fma.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libsepaihrd_${NAME}.so /tmp/kv_${NAME}_strict.o /tmp/kv_${NAME}_fma.o kernels_f32.o ensemble.o sampler.o capi.o
