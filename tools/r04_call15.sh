#!/bin/bash
# round 4, GPU call 15: this round's kernels (with the builds for fields without flat cells) against round 3's own tree (build/r3src), order alternating
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c15
abx() {
  echo "### $*"
  echo -n "round 4      : "; python3 tools/bench_line.py "$@" --parity-stride 0
  echo -n "round 3 HEAD : "; (cd build/r3src && python3 tools/bench_line.py "$@" --parity-stride 0)
  echo -n "round 4      : "; python3 tools/bench_line.py "$@" --parity-stride 0
  echo -n "round 3 HEAD : "; (cd build/r3src && python3 tools/bench_line.py "$@" --parity-stride 0)
}
{
abx --record none --steps 10
abx --steps 10
abx --scenario fisheye --record none --steps 10
abx --scenario fisheye --steps 10
abx --dtype f32 --rays 8388608 --record none --steps 5
abx --scenario interface --record none --steps 5
abx --method 1 --record none --steps 5
abx --method 8 --record none --steps 5
} > gpurun_out/r4_c15/ab.txt 2>&1
cat gpurun_out/r4_c15/ab.txt | awk -F'  +' '{print $1" | "$2" | "$3}' | cut -c1-170
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "not full_1m and not 8m and not headline" > gpurun_out/r4_c15/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_c15/pytest.log
