#!/bin/bash
# round 4, GPU call 36: full GPU suite on the build with the grouped fallback, then the reference-order kernels' profiles, rates, sweep
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c36
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r4_c36/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_c36/pytest.log
bash tools/r04_profile_all.sh 2 > gpurun_out/r4_c36/profile2.log 2>&1
echo "profiles done"
bash tools/all_methods_rate.sh > gpurun_out/r4_c36/all_methods_rate.txt 2>&1
echo "rates done"
timeout -k 10 600 python3 tools/parity_sweep.py > gpurun_out/r4_c36/parity_sweep.txt 2>&1
tail -2 gpurun_out/r4_c36/parity_sweep.txt | cut -c1-400
