#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c40
timeout -k 10 1000 python3 tools/parity_sweep_1m.py > gpurun_out/r4_c40/parity_sweep_1m.txt 2>&1; echo "sweep rc=$?"; tail -50 gpurun_out/r4_c40/parity_sweep_1m.txt | cut -c1-200
