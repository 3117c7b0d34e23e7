#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c46
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r4_c46/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4_c46/pytest.log
bash tools/all_methods_rate.sh > gpurun_out/r4_c46/all_methods_rate.txt 2>&1
grep -E "fisheye.*method (3|4|5|7|9) " gpurun_out/r4_c46/all_methods_rate.txt | cut -c1-200
tools/profile_config.sh r04q_fisheye_op9_none --scenario fisheye --method 9 --rays 524288 --record none > gpurun_out/r4_c46/profile.log 2>&1
tools/profile_config.sh r04q_fisheye_op5_none --scenario fisheye --method 5 --rays 524288 --record none >> gpurun_out/r4_c46/profile.log 2>&1
echo done
