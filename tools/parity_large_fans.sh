#!/bin/bash
# Parity of the default forms of op1/2/6/7/8 (op7 with --fast-field too) on LARGE fans: 65 536 rays per scenario with the full record, every 4th ray -- 16 384
# rays, every recorded row -- against the oracle (bench.py's parity_check: per-quantity relative error, step counts exactly).
set -u
for scen in vert_heterogeneous fisheye interface; do
  rows=""; [ "$scen" = interface ] && rows="--rec-rows 9000"
  for m in 1 2 6 7 8 "7 --fast-field"; do
    python3 bench.py --scenario $scen --method $m --rays 65536 --record full $rows --steps 2 --cpu-seconds 0 --parity-stride 4 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['parity_check']
print(f\"{sys.argv[1]:19s} op{' '.join(sys.argv[2:]):15s} rays {p['rays']:6d}  rows {p['rows_compared']:5d}  steps equal {p['steps_equal']}  final {p['max_rel_err']:.1e}  rows {p['rows_max_rel_err']:.1e}  ok {p['ok']}\")" $scen $m
  done
done
