#!/bin/bash
# round 4, GPU call 30: A/B of the wave-uniform window for the reference-order lookup (build/ab/librtmi_uni.so) against the round's build
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c30
RTMI_LIB_PATH=build/ab/librtmi_uni.so timeout -k 10 600 python3 -m pytest tests/test_gpu_exact.py -m gpu -q -x > gpurun_out/r4_c30/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_c30/pytest.log
{
echo "## uni: a wave whose live lanes share one cell reads the reference-order lookup's window, knots and reciprocals through the scalar cache"
for args in "--method 7 --record none --steps 3" "--method 3 --record none --steps 3" "--method 4 --record none --steps 3" \
  "--method 6 --reference-order --record none --steps 3" "--method 9 --rays 524288 --record none --steps 3" "--method 5 --rays 524288 --record none --steps 3" \
  "--scenario anisotropy --record none --steps 3" "--scenario anisotropy --method 10 --rays 524288 --record none --steps 3" \
  "--scenario interface --method 7 --record none --steps 3" "--scenario interface --method 9 --rays 524288 --record none --steps 3" "--scenario interface --method 4 --record none --steps 3" \
  "--scenario fisheye --method 7 --record none --steps 3" "--scenario fisheye --method 3 --record none --steps 3" \
  "--method 7 --rays 65536 --record none --steps 5" "--method 3 --steps 3"; do
  bash tools/ab_variants.sh "$args" build/ab/librtmi_base.so build/ab/librtmi_uni.so
done
} > gpurun_out/r4_c30/ab.txt 2>&1
cat gpurun_out/r4_c30/ab.txt
