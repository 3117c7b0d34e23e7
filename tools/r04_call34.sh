#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c34
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r4_c34/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4_c34/pytest.log
