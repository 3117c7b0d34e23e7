#!/bin/bash
# round 4, GPU call 53: the headline under auto / sliced / plain, several processes each: does RTMI_LAUNCH_AUTO's exploration pick the right schedule?
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c53
{
for i in 1 2 3; do
  for m in auto sliced plain; do
    echo -n "$m : "; RTMI_DEBUG_AUTO=1 python3 tools/bench_line.py --steps 20 --mode $m
  done
done
} > gpurun_out/r4_c53/modes.txt 2>&1
cut -c1-200 gpurun_out/r4_c53/modes.txt
