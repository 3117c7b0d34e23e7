#!/bin/bash
# round 4, GPU call 35: the per-lane fallback of the reference-order lookup in 2 / 4 groups of rows (register count) -- A/B against the round's build
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c35
RTMI_LIB_PATH=build/ab/librtmi_ph2.so timeout -k 10 600 python3 -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -m gpu -q -x -k "exact or window or tile or cfg5 or critical" > gpurun_out/r4_c35/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4_c35/pytest.log
{
echo "## ph2 / ph4: the per-lane fallback of rt::ex::n_gradient reads its 4 x 4 window in 2 / 4 groups of rows (base: all at once)"
for args in "--method 3 --record none --steps 3" "--method 4 --record none --steps 3" "--method 7 --record none --steps 3" "--method 5 --rays 524288 --record none --steps 3" "--method 9 --rays 524288 --record none --steps 3" \
  "--scenario anisotropy --record none --steps 3" "--scenario anisotropy --method 10 --rays 524288 --record none --steps 3" \
  "--scenario interface --method 9 --rays 524288 --record none --steps 3" "--scenario interface --method 5 --rays 524288 --record none --steps 3" "--scenario interface --method 3 --record none --steps 3" \
  "--scenario fisheye --method 3 --record none --steps 3" "--method 3 --order shuffled --record none --steps 3" "--method 9 --rays 262144 --order shuffled --record none --steps 3" \
  "--method 3 --rays 65536 --record none --steps 5" "--scenario anisotropy --record none --steps 5 --total-rays 1048576 --emulate-world 8" "--method 3 --steps 3"; do
  bash tools/ab_variants.sh "$args" build/ab/librtmi_base.so build/ab/librtmi_ph2.so build/ab/librtmi_ph4.so
done
} > gpurun_out/r4_c35/ab.txt 2>&1
cat gpurun_out/r4_c35/ab.txt | cut -c1-215
