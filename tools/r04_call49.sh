#!/bin/bash
# round 4, GPU call 49: fisheye x op9, window forced on / per-lane, by batch size (waves per scalar cache): is the wait a capacity effect?
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c49
{
for r in 32768 65536 131072 262144 524288; do
  for fp in window global; do
    python3 tools/bench_line.py --scenario fisheye --method 9 --rays $r --record none --steps 3 --field-path $fp
  done
done
for r in 65536 262144 1048576; do
  for fp in window global; do
    python3 tools/bench_line.py --scenario fisheye --method 3 --rays $r --record none --steps 3 --field-path $fp
  done
done
} > gpurun_out/r4_c49/by_size.txt 2>&1
cut -c1-200 gpurun_out/r4_c49/by_size.txt
