#!/bin/bash
# round 4, GPU call 55: RTMI_LAUNCH_AUTO with each schedule's second time: which schedule the BASELINE configurations keep, three processes each
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c55
{
for i in 1 2 3; do
  python3 tools/bench_line.py --steps 20 | sed 's/^/headline auto: /'
done
for a in "--record none --steps 5" "--scenario fisheye --record none --steps 5" "--scenario fisheye --record full --steps 5" "--scenario anisotropy --record none --steps 3" "--scenario interface --record none --steps 5" "--method 9 --rays 524288 --record none --steps 3"; do
  for m in auto sliced plain; do echo -n "$m : "; python3 tools/bench_line.py $a --mode $m; done
done
python3 bench.py > gpurun_out/r4_c55/bench_default.json 2>/dev/null; cut -c1-160 gpurun_out/r4_c55/bench_default.json
} > gpurun_out/r4_c55/auto.txt 2>&1
cut -c1-200 gpurun_out/r4_c55/auto.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4_c55/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4_c55/pytest.log
