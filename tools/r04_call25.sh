#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c25
for m in 1 8 6 7 2; do
python3 bench.py --scenario interface --method $m --rays 1048576 --record stride:16 --rec-rows 600 --steps 2 --cpu-seconds 0 --parity-stride 64 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['parity_check']
print(f\"interface 1 048 576 rays op{sys.argv[1]}  checked {p['rays']:6d} rays x {p['rows_compared']} rows (every 16th)  steps equal {p['steps_equal']}  final {p['max_rel_err']:.1e}  rows {p['rows_max_rel_err']:.1e}  ok {p['ok']}\")" $m
done > gpurun_out/r4_c25/iface_1m_parity.txt 2>&1
cat gpurun_out/r4_c25/iface_1m_parity.txt
