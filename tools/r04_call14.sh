#!/bin/bash
# round 4, GPU call 14: this round's kernels against round 3's (its own tree under build/r3src) and against a build with the flat-cell map's tests compiled out
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c14
ROOT=$PWD
abx() {
  echo "### $*"
  for round in 1 2; do
    echo -n "round 3 HEAD      : "; (cd build/r3src && python3 tools/bench_line.py "$@" --parity-stride 0)
    echo -n "round 4           : "; python3 tools/bench_line.py "$@" --parity-stride 0
    echo -n "round 4, no tests : "; RTMI_LIB_PATH=$ROOT/build/ab/librtmi_noflatcode.so python3 tools/bench_line.py "$@" --parity-stride 0
  done
}
{
abx --record none --steps 10
abx --steps 10
abx --scenario fisheye --record none --steps 10
abx --scenario fisheye --steps 10
abx --dtype f32 --rays 8388608 --record none --steps 5
abx --rays 65536 --record none --steps 20
abx --scenario interface --record none --steps 5
} > gpurun_out/r4_c14/ab.txt 2>&1
cat gpurun_out/r4_c14/ab.txt | awk -F'  +' '{print $1" | "$2" | "$3}' | cut -c1-170
