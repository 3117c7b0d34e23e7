#!/bin/bash
# round 4, GPU call 50: the window attempt leaves on the lanes' own estimates before anything is loaded (ff) -- the fisheye cliff, and the coherent fans
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c50
RTMI_LIB_PATH=build/ab/librtmi_ff.so RTMI_WINDOW_MIN_RAYS=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -m gpu -q -x -k "exact or window or tile or cfg5 or critical or golden or aniso" > gpurun_out/r4_c50/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4_c50/pytest.log
{
echo "## ff: the window attempt is abandoned on the lanes' own cell estimates (a ballot) before any scalar load; --field-path window forces the window on, global is the per-lane gather"
for args in "--scenario fisheye --method 9 --rays 524288 --record none --steps 3 --field-path window" "--scenario fisheye --method 9 --rays 524288 --record none --steps 3 --field-path global" \
  "--scenario fisheye --method 5 --rays 524288 --record none --steps 3 --field-path window" "--scenario fisheye --method 5 --rays 524288 --record none --steps 3 --field-path global" \
  "--scenario fisheye --method 9 --rays 65536 --record none --steps 3 --field-path window" \
  "--scenario fisheye --method 3 --record none --steps 3" "--scenario fisheye --method 7 --record none --steps 3" \
  "--method 7 --record none --steps 3" "--method 3 --record none --steps 3" "--method 9 --rays 524288 --record none --steps 3" "--method 5 --rays 524288 --record none --steps 3" \
  "--scenario interface --method 9 --rays 524288 --record none --steps 3" "--scenario interface --method 5 --rays 524288 --record none --steps 3" "--scenario interface --method 7 --record none --steps 3" \
  "--method 3 --order shuffled --record none --steps 3"; do
  bash tools/ab_variants.sh "$args" build/ab/librtmi_base.so build/ab/librtmi_ff.so
done
} > gpurun_out/r4_c50/ab.txt 2>&1
cat gpurun_out/r4_c50/ab.txt | cut -c1-215
