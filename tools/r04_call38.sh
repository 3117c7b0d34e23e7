#!/bin/bash
# round 4, GPU call 38: golden-section searches with the reference's trip count as a scalar loop count (gfix) -- A/B against the round's build
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c38
RTMI_LIB_PATH=build/ab/librtmi_gfix.so timeout -k 10 600 python3 -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -m gpu -q -x -k "exact or window or tile or cfg5 or critical or golden or aniso" > gpurun_out/r4_c38/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4_c38/pytest.log
{
echo "## gfix: the golden-section loops run the reference's 37 iterations as a scalar count (checked per lane afterwards), rare branches behind a ballot"
for args in "--scenario anisotropy --record none --steps 3" "--scenario anisotropy --method 10 --rays 524288 --record none --steps 3" \
  "--method 9 --rays 524288 --record none --steps 3" "--method 5 --rays 524288 --record none --steps 3" \
  "--scenario interface --method 9 --rays 524288 --record none --steps 3" "--scenario interface --method 5 --rays 524288 --record none --steps 3" \
  "--scenario fisheye --method 9 --rays 524288 --record none --steps 3" "--scenario fisheye --method 5 --rays 524288 --record none --steps 3" \
  "--scenario anisotropy --record none --steps 5 --total-rays 1048576 --emulate-world 8" "--method 9 --rays 65536 --record none --steps 5"; do
  bash tools/ab_variants.sh "$args" build/ab/librtmi_base.so build/ab/librtmi_gfix.so
done
} > gpurun_out/r4_c38/ab.txt 2>&1
cat gpurun_out/r4_c38/ab.txt | cut -c1-215
