#!/bin/bash
# round 4, GPU call 56: the first half of the profile set (the fast-form kernels) again, from the final build
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c56
bash tools/r04_profile_all.sh 1 > gpurun_out/r4_c56/profile1.log 2>&1
echo "profiles done"
