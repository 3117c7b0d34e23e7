#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c43
tools/profile_config.sh r04q_fisheye_op9_none --scenario fisheye --method 9 --rays 524288 --record none > gpurun_out/r4_c43/profile.log 2>&1
tools/profile_config.sh r04q_fisheye_op5_none --scenario fisheye --method 5 --rays 524288 --record none >> gpurun_out/r4_c43/profile.log 2>&1
echo done
