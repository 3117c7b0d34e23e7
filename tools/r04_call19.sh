#!/bin/bash
# round 4: the default bench line as the driver runs it
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c19
start=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_c19/bench_default.json 2> gpurun_out/r4_c19/bench_default.err; echo "bench rc=$?"
end=$(date +%s.%N); echo "wall $(echo "$end - $start" | bc) s"
python3 -c "
import json; d=json.loads(open('gpurun_out/r4_c19/bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['bound'], r['traffic'], r.get('traffic_source'), r.get('traffic_stale'), r['kernel'], r['vgprs'], d['cpu_baseline']['value'], d['parity_check'])"
