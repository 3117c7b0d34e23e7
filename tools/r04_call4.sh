#!/bin/bash
# round 4, GPU call 4: the flat-cell map -- full suite, then A/B against RTMI_NO_FLAT=1 in one session (two interleaved rounds)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c4
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4_c4/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r4_c4/pytest.log
RTMI_DEBUG=1 python3 tools/bench_line.py --scenario interface --record none --steps 1 --rays 65536 2>&1 | grep -i "flat" | head -3
ab() {
  echo "### $*"
  for round in 1 2; do
    echo -n "no map   : "; RTMI_NO_FLAT=1 python3 tools/bench_line.py "$@"
    echo -n "flat map : "; python3 tools/bench_line.py "$@"
  done
}
{
ab --scenario interface --record none --steps 5
ab --scenario interface --steps 5 --rec-rows 4100
ab --scenario interface --method 2 --record none --steps 5
ab --scenario interface --method 8 --record none --steps 5
ab --steps 10
ab --record none --steps 10
ab --scenario fisheye --record none --steps 10
ab --dtype f32 --rays 8388608 --record none --steps 5
ab --rays 65536 --record none --steps 20
} > gpurun_out/r4_c4/ab_flat_map.txt 2>&1
cat gpurun_out/r4_c4/ab_flat_map.txt
