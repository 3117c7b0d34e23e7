#!/bin/bash
# round 4, GPU call 32: the wave-uniform window gated by batch size (FieldDev::window): full GPU suite, A/B by RTMI_WINDOW_MIN_RAYS, sweep
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c32
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r4_c32/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_c32/pytest.log
{
echo "## the round's build: default gate (>= 196 608 rays) | window always (RTMI_WINDOW_MIN_RAYS=0) | window never (=99999999); two interleaved rounds"
for args in "--method 7 --record none --steps 3" "--method 3 --steps 3" "--method 6 --reference-order --record none --steps 3" "--scenario anisotropy --record none --steps 3" \
  "--scenario interface --method 9 --rays 524288 --record none --steps 3" "--scenario interface --method 5 --rays 524288 --record none --steps 3" \
  "--method 7 --rays 262144 --record none --steps 5" "--method 7 --rays 131072 --record none --steps 5" "--method 7 --rays 65536 --record none --steps 5" \
  "--scenario anisotropy --record none --steps 5 --total-rays 1048576 --emulate-world 8" "--method 3 --rays 4096 --record none --steps 5"; do
  for round in 1 2; do
    for w in default 0 99999999; do
      echo -n "window $w : "
      if [ $w = default ]; then python3 tools/bench_line.py $args; else RTMI_WINDOW_MIN_RAYS=$w python3 tools/bench_line.py $args; fi
    done
  done
done
} > gpurun_out/r4_c32/ab.txt 2>&1
cat gpurun_out/r4_c32/ab.txt | cut -c1-200
timeout -k 10 600 python3 tools/parity_sweep.py > gpurun_out/r4_c32/parity_sweep.txt 2>&1
tail -2 gpurun_out/r4_c32/parity_sweep.txt | cut -c1-400
