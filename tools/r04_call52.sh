#!/bin/bash
# round 4, GPU call 52: the final build -- GPU suite (default gate, and the window forced on), the reference-order kernels' profiles, rates, both sweeps, the default line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c52
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r4_c52/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_c52/pytest.log
RTMI_WINDOW_MIN_RAYS=0 timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r4_c52/pytest_window_always.log 2>&1; echo "pytest (window always) rc=$?"; tail -2 gpurun_out/r4_c52/pytest_window_always.log
bash tools/r04_profile_all.sh 2 > gpurun_out/r4_c52/profile2.log 2>&1
echo "profiles done"
bash tools/all_methods_rate.sh > gpurun_out/r4_c52/all_methods_rate.txt 2>&1
echo "rates done"
timeout -k 10 600 python3 tools/parity_sweep.py > gpurun_out/r4_c52/parity_sweep.txt 2>&1
tail -1 gpurun_out/r4_c52/parity_sweep.txt | cut -c1-400
timeout -k 10 600 python3 tools/parity_sweep_1m.py > gpurun_out/r4_c52/parity_sweep_1m.txt 2>&1
tail -1 gpurun_out/r4_c52/parity_sweep_1m.txt | cut -c1-400
python3 bench.py > gpurun_out/r4_c52/bench_default.json 2> gpurun_out/r4_c52/bench_default.err; cut -c1-200 gpurun_out/r4_c52/bench_default.json
