#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c29
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_exact.py -m gpu -q -s -k "critical_rays or op7" > gpurun_out/r4_c29/pytest.log 2>&1; echo "pytest rc=$?"; grep "^op\|passed\|failed\|vs oracle" gpurun_out/r4_c29/pytest.log
