#!/bin/bash
# reference-order methods: global gathers (auto) against the LDS tile (field_path 2), same session
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c23
{
for cfg in "--method 3 --record none --steps 3" "--method 9 --rays 524288 --record none --steps 3" "--scenario anisotropy --record none --steps 3" "--scenario interface --method 9 --rays 524288 --record none --steps 3" "--scenario interface --method 4 --record none --steps 3" "--method 7 --reference-order --record none --steps 3" "--scenario fisheye --method 3 --record none --steps 3"; do
  for round in 1 2; do
    echo -n "global : "; python3 tools/bench_line.py $cfg
    echo -n "tile   : "; python3 tools/bench_line.py $cfg --field-path lds
  done
done
} > gpurun_out/r4_c23/ab.txt 2>&1
cat gpurun_out/r4_c23/ab.txt | awk -F'  +' '{print $1" | "$2" | "$3}' | cut -c1-170
