#!/bin/bash
# Register / spill / scratch figures of the evaluation kernels from a device-only compile (no GPU needed).
#   tools/kernel_regs.sh fma|strict [extra -D flags...]      prints one line per kernel whose name matches $PATTERN (default: eval)
A=${1:-fma}; shift
C=mathematical-modeling-of-infectious-diseases-v1_amd/csrc
F="-ffp-contract=fast -DSEPAIHRD_ARITH_FMA=1"; [ "$A" = strict ] && F="-ffp-contract=off -DSEPAIHRD_ARITH_FMA=0"
OUT=${OUT:-/tmp/kernel_regs_$A.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Iinclude -I$C -Wno-unused-function $F "$@" --cuda-device-only -S $C/sepaihrd_kernels.hip -o $OUT || exit 1
python3 - "$OUT" "${PATTERN:-eval}" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:\s+(\d+)\n(.*?)\.wavefront_size", txt, re.S):
    agpr, body = m.group(1), m.group(2)
    name = re.search(r"\.name:\s+(\S+)", body).group(1)
    if sys.argv[2] not in name: continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, body).group(1)
    short = re.sub(r"_ZN8sepaihrd12_GLOBAL__N_1\d+", "", name).split("EEvNS")[0]
    print("%-52s vgpr %3s (agpr %3s, spilled %2s) sgpr %3s (spilled %2s) scratch %4s lds %s" % (short, g("vgpr_count"), agpr, g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
PY
