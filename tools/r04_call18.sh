#!/bin/bash
# round 4, GPU call 18: profile set half 2, every method's rate, the parity sweep
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c18
bash tools/r04_profile_all.sh 2 > gpurun_out/r4_c18/profile2.log 2>&1
echo "profiles done"
bash tools/all_methods_rate.sh > gpurun_out/r4_c18/all_methods_rate.txt 2>&1
echo "rates done"
timeout -k 10 900 python3 tools/parity_sweep.py > gpurun_out/r4_c18/parity_sweep.txt 2>&1
tail -3 gpurun_out/r4_c18/parity_sweep.txt
