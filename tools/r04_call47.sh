#!/bin/bash
# round 4, GPU call 47: final artifacts -- smoke(), the parity matrix at north-star size (its ms/pass column), the critical-ray window, the default line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c47
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_c47/smoke.log 2>&1; tail -2 gpurun_out/r4_c47/smoke.log
timeout -k 10 600 python3 tools/parity_sweep_1m.py > gpurun_out/r4_c47/parity_sweep_1m.txt 2>&1; tail -1 gpurun_out/r4_c47/parity_sweep_1m.txt | cut -c1-300
timeout -k 10 600 python3 tools/critical_ray_window.py > gpurun_out/r4_c47/critical_ray_window.txt 2>&1; grep -c "reference" gpurun_out/r4_c47/critical_ray_window.txt
python3 bench.py > gpurun_out/r4_c47/bench_default.json 2> gpurun_out/r4_c47/bench_default.err; cut -c1-200 gpurun_out/r4_c47/bench_default.json
python3 bench.py --gpus 2 --backend gloo --all-on-device 0 --rays 65536 --steps 2 --cpu-seconds 0 > gpurun_out/r4_c47/bench_2ranks.json 2> gpurun_out/r4_c47/bench_2ranks.err; echo "2-rank rc=$?"; cut -c1-200 gpurun_out/r4_c47/bench_2ranks.json
