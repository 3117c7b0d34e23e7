#!/bin/bash
# round 4, GPU call 33: profiles of the kernels the wave-uniform window changed (the reference-order methods), every method's rate, the default line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c33
timeout -k 10 300 python3 -m pytest tests/test_gpu_exact.py -m gpu -q -x > gpurun_out/r4_c33/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/r4_c33/pytest.log
bash tools/r04_profile_all.sh 2 > gpurun_out/r4_c33/profile2.log 2>&1
echo "profiles done"
bash tools/all_methods_rate.sh > gpurun_out/r4_c33/all_methods_rate.txt 2>&1
echo "rates done"
python3 bench.py > gpurun_out/r4_c33/bench_default.json 2> gpurun_out/r4_c33/bench_default.err; cut -c1-300 gpurun_out/r4_c33/bench_default.json
