#!/bin/bash
# round 5, GPU call 40: fewer rays per re-trace wave (a lone wave issues every path any of its lanes takes): 64 / 32 / 16 / 8 lanes per block
O=gpurun_out/r5_c40; mkdir -p $O
{
for m in 8 1 6; do
for lanes in 64 32 16 8; do
echo -n "lanes $lanes: "; env RTMI_RETRACE_LANES=$lanes python tools/bench_line.py --scenario interface --method $m --record none --steps 10 --mode plain
done
done
} 2>&1 | tee $O/times.txt
