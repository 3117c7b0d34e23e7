#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c44
{
for args in "--scenario fisheye --method 9 --rays 524288 --record none --steps 3" "--scenario fisheye --method 5 --rays 524288 --record none --steps 3" \
  "--scenario fisheye --method 3 --record none --steps 3" "--scenario fisheye --method 4 --record none --steps 3" "--scenario fisheye --method 7 --record none --steps 3" \
  "--scenario fisheye --method 9 --rays 524288 --record none --steps 3 --mode plain"; do
  for w in default 99999999; do
    echo -n "window $w : "
    if [ $w = default ]; then python3 tools/bench_line.py $args; else RTMI_WINDOW_MIN_RAYS=$w python3 tools/bench_line.py $args; fi
  done
done
} > gpurun_out/r4_c44/ab.txt 2>&1
cat gpurun_out/r4_c44/ab.txt | cut -c1-200
