#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c13
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4_c13/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r4_c13/pytest.log
timeout -k 10 900 python3 bench.py --scenario interface --method 7 --rays 65536 --record full --rec-rows 9000 --steps 2 --cpu-seconds 0 --parity-stride 8 > gpurun_out/r4_c13/iface_op7_65536_parity.json 2> gpurun_out/r4_c13/iface_op7.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r4_c13/iface_op7_65536_parity.json').read().strip().splitlines()[-1]); print(d['value'], d['parity_check'])"
timeout -k 10 600 python3 tools/parity_sweep.py --methods 7 > gpurun_out/r4_c13/sweep_op7.txt 2>&1; grep "interface\|^#" gpurun_out/r4_c13/sweep_op7.txt
