#!/bin/bash
# A/B of librtmi builds in one session: tools/ab_variants.sh "<bench args>" lib1.so lib2.so ...  (two interleaved rounds)
args="$1"; shift
for round in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib) : "
    RTMI_LIB_PATH=$lib python tools/bench_line.py $args
  done
done
