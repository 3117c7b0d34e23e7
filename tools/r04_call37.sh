#!/bin/bash
# round 4, GPU call 37: the whole GPU suite with the scalar-cache window forced on for every batch size (RTMI_WINDOW_MIN_RAYS=0), then 1 M-ray
# parity of reference-order methods (window on by default there) against the oracle
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c37
RTMI_WINDOW_MIN_RAYS=0 timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r4_c37/pytest_window_always.log 2>&1; echo "pytest (window always) rc=$?"; tail -3 gpurun_out/r4_c37/pytest_window_always.log
for a in "--scenario interface --method 7 --rays 1048576 --record stride:16 --rec-rows 600 --parity-stride 64" "--method 3 --rays 1048576 --record stride:16 --parity-stride 256" \
  "--method 9 --rays 1048576 --record stride:16 --parity-stride 512" "--scenario fisheye --method 4 --rays 1048576 --record stride:16 --parity-stride 256" \
  "--scenario interface --method 5 --rays 1048576 --record stride:16 --rec-rows 600 --parity-stride 512" "--method 6 --reference-order --rays 1048576 --record stride:16 --parity-stride 256"; do
  python3 bench.py $a --steps 2 --cpu-seconds 0 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['parity_check']
print(f\"{' '.join(sys.argv[1:]):110s} checked {p['rays']:6d} rays x {p['rows_compared']} rows  steps equal {p['steps_equal']}  final {p['max_rel_err']:.1e}  rows {p['rows_max_rel_err']:.1e}  ok {p['ok']}  {d['ms_per_step']:.1f} ms/pass\")" $a
done > gpurun_out/r4_c37/parity_1m_reference_order.txt 2>&1
cat gpurun_out/r4_c37/parity_1m_reference_order.txt
