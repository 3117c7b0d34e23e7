#!/bin/bash
# round 4, GPU call 5: rtmi_shard tests; profiles of the kernels this round changed (interface op6 with the flat-cell map, interface op9 with the inline tie path, op7's default)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c5
timeout -k 10 600 python3 -m pytest tests/test_gpu_shard.py tests/test_cabi_native.py -m gpu -q > gpurun_out/r4_c5/pytest.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r4_c5/pytest.log
tools/profile_config.sh r04_iface_none --scenario interface --record none
tools/profile_config.sh r04_iface_op9 --scenario interface --method 9 --rays 524288 --record none
tools/profile_config.sh r04_vert_op7_none --method 7 --record none
echo done
