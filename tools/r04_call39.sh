#!/bin/bash
# round 4, GPU call 39: the parity matrix at north-star size; the sliced schedule's slice length for cfg5
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c39
timeout -k 10 900 python3 tools/parity_sweep_1m.py > gpurun_out/r4_c39/parity_sweep_1m.txt 2>&1; echo "sweep rc=$?"; tail -45 gpurun_out/r4_c39/parity_sweep_1m.txt | cut -c1-200
{
for args in "--scenario anisotropy --record none --steps 3" "--scenario anisotropy --record none --steps 3 --slice-steps 256" "--scenario anisotropy --record none --steps 3 --slice-steps 1024" \
  "--scenario anisotropy --record none --steps 3 --slice-steps 128" "--scenario anisotropy --record none --steps 3 --mode plain" "--scenario anisotropy --record none --steps 3 --mode refill" \
  "--scenario anisotropy --record none --steps 3 --mode plain --block 128" "--scenario anisotropy --record none --steps 3 --mode plain --block 512"; do
  python3 tools/bench_line.py $args
done
} > gpurun_out/r4_c39/cfg5_schedules.txt 2>&1
cat gpurun_out/r4_c39/cfg5_schedules.txt | cut -c1-200
