#!/bin/bash
# round 4, GPU call 51: the cache-line touch with 18 DISTINCT scratch scalar registers (touch2): fisheye x op9 / op5 with the window forced on
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c51
{
echo "## touch2: as touch (call 45), every touch load into a scalar register of its own"
for args in "--scenario fisheye --method 9 --rays 524288 --record none --steps 3 --field-path window" "--scenario fisheye --method 5 --rays 524288 --record none --steps 3 --field-path window" \
  "--scenario fisheye --method 9 --rays 65536 --record none --steps 3 --field-path window" \
  "--method 9 --rays 524288 --record none --steps 3" "--scenario interface --method 9 --rays 524288 --record none --steps 3" "--scenario interface --method 5 --rays 524288 --record none --steps 3"; do
  bash tools/ab_variants.sh "$args" build/ab/librtmi_base.so build/ab/librtmi_touch2.so
done
} > gpurun_out/r4_c51/ab.txt 2>&1
cat gpurun_out/r4_c51/ab.txt | cut -c1-215
