#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c28
timeout -k 10 900 python3 tools/critical_ray_window.py > gpurun_out/r4_c28/critical_ray_window.txt 2>&1; cat gpurun_out/r4_c28/critical_ray_window.txt
