#!/bin/bash
# round 4, GPU call 45: the window's cache-line touch for the golden-section kernels (touch) -- A/B: the fisheye cliff, and what it costs the coherent fans
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c45
RTMI_LIB_PATH=build/ab/librtmi_touch.so RTMI_WINDOW_MIN_RAYS=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -m gpu -q -x -k "exact or window or tile or cfg5 or critical or golden or aniso" > gpurun_out/r4_c45/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4_c45/pytest.log
{
echo "## touch: the golden-section kernels ask for one dword of each of the window's 18 cache lines at once before using any (the compiler loads the window piece by piece there)"
for args in "--scenario fisheye --method 9 --rays 524288 --record none --steps 3" "--scenario fisheye --method 5 --rays 524288 --record none --steps 3" \
  "--method 9 --rays 524288 --record none --steps 3" "--method 5 --rays 524288 --record none --steps 3" "--scenario anisotropy --record none --steps 3" "--scenario anisotropy --method 10 --rays 524288 --record none --steps 3" \
  "--scenario interface --method 9 --rays 524288 --record none --steps 3" "--scenario interface --method 5 --rays 524288 --record none --steps 3" \
  "--method 9 --rays 524288 --order shuffled --record none --steps 3" "--scenario anisotropy --record none --steps 5 --total-rays 1048576 --emulate-world 8"; do
  bash tools/ab_variants.sh "$args" build/ab/librtmi_base.so build/ab/librtmi_touch.so
done
echo "## the other reference-order methods on fisheye, window on (default) / off (RTMI_WINDOW_MIN_RAYS=99999999), base build: do they batch their loads?"
for args in "--scenario fisheye --method 3 --record none --steps 3" "--scenario fisheye --method 6 --reference-order --record none --steps 3" "--scenario fisheye --method 1 --reference-order --record none --steps 3"; do
  for w in default 99999999; do
    echo -n "window $w : "
    if [ $w = default ]; then python3 tools/bench_line.py $args; else RTMI_WINDOW_MIN_RAYS=$w python3 tools/bench_line.py $args; fi
  done
done
} > gpurun_out/r4_c45/ab.txt 2>&1
cat gpurun_out/r4_c45/ab.txt | cut -c1-215
