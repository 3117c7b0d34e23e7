#!/bin/bash
# round 4, GPU call 41: the wave-uniform window with the cell ESTIMATED and checked against the table's knots instead of searched (est) -- A/B
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c41
RTMI_LIB_PATH=build/ab/librtmi_est.so RTMI_WINDOW_MIN_RAYS=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -m gpu -q -x -k "exact or window or tile or cfg5 or critical or golden or aniso or edge or clamp" > gpurun_out/r4_c41/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4_c41/pytest.log
{
echo "## est: the window path takes the first live lane's ESTIMATE of its cell and checks every live lane against the table entry's knots (no interval search)"
for args in "--method 7 --record none --steps 3" "--method 3 --record none --steps 3" "--method 4 --record none --steps 3" "--method 6 --reference-order --record none --steps 3" \
  "--method 9 --rays 524288 --record none --steps 3" "--scenario anisotropy --record none --steps 3" \
  "--scenario interface --method 7 --record none --steps 3" "--scenario interface --method 9 --rays 524288 --record none --steps 3" "--scenario interface --method 3 --record none --steps 3" \
  "--scenario fisheye --method 7 --record none --steps 3" "--scenario fisheye --method 3 --record none --steps 3" "--method 7 --rays 131072 --record none --steps 5" "--method 3 --steps 3"; do
  bash tools/ab_variants.sh "$args" build/ab/librtmi_base.so build/ab/librtmi_est.so
done
} > gpurun_out/r4_c41/ab.txt 2>&1
cat gpurun_out/r4_c41/ab.txt | cut -c1-215
