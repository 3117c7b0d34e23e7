#!/bin/bash
# Shader clock and package power while a bench configuration runs: tools/clock_sample.sh <tag> <bench args...>
# (rocm-smi sampled every 0.25 s in the background; samples to gpurun_out/clk_<tag>.log)
tag=$1; shift
mkdir -p gpurun_out
( for i in $(seq 1 40); do rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clk_$tag.log &
smi=$!
python3 tools/bench_line.py --steps 1000 --warmup 20 "$@"
kill $smi 2>/dev/null; wait $smi 2>/dev/null
grep -c sclk gpurun_out/clk_$tag.log
