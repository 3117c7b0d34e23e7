#!/bin/bash
# round 4, GPU call 22: the final build -- full GPU suite, then the profile set's second half again (its kernels changed: LDS sin/cos table), every method's rate, the op7 sweep
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c22
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r4_c22/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_c22/pytest.log
bash tools/r04_profile_all.sh 2 > gpurun_out/r4_c22/profile2.log 2>&1
echo "profiles done"
bash tools/all_methods_rate.sh > gpurun_out/r4_c22/all_methods_rate.txt 2>&1
echo "rates done"
timeout -k 10 600 python3 tools/parity_sweep.py > gpurun_out/r4_c22/parity_sweep.txt 2>&1
tail -2 gpurun_out/r4_c22/parity_sweep.txt | cut -c1-300
