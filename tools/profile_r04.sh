#!/bin/bash
# Round-4 profiles of the shipped library, every BASELINE config (about ten minutes of one MI355X):
#   /usr/local/graft/bin/gpurun --timeout 1150 -- 'bash tools/profile_r04.sh'    then    for t in ...; python3 tools/summarize_profile.py $t
set -o pipefail
bash tools/profile_round.sh r04_c1_fma "--steps 20 --warmup 3" || exit 1
bash tools/profile_round.sh r04_c1_strict "--steps 20 --warmup 3 --arith strict" || exit 1
bash tools/profile_round.sh r04_c3_fma "--workload c3 --steps 10 --warmup 3" || exit 1
bash tools/profile_round.sh r04_c2_fma "--workload c2 --steps 10 --warmup 3" || exit 1
bash tools/profile_round.sh r04_c5_fma "--workload c5 --steps 5 --warmup 2" || exit 1
bash tools/profile_round.sh r04_c5_f32 "--workload c5 --precision f32 --steps 5 --warmup 2" || exit 1
