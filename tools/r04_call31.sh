#!/bin/bash
# round 4, GPU call 31: the wave-uniform window at the latency-bound sizes (few waves per SIMD)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c31
{
for args in "--scenario anisotropy --record none --steps 5 --total-rays 1048576 --emulate-world 8" "--method 7 --rays 131072 --record none --steps 5" "--method 7 --rays 262144 --record none --steps 5" \
  "--method 3 --rays 65536 --record none --steps 5" "--method 3 --rays 131072 --record none --steps 5" "--method 9 --rays 65536 --record none --steps 5" "--method 7 --rays 16384 --record none --steps 5" \
  "--scenario interface --method 3 --rays 65536 --record none --steps 5"; do
  bash tools/ab_variants.sh "$args" build/ab/librtmi_base.so build/ab/librtmi_uni.so
done
} > gpurun_out/r4_c31/ab.txt 2>&1
cat gpurun_out/r4_c31/ab.txt
