#!/bin/bash
# round 5, GPU call 53: the small parity sweep (4 096-ray fans and 2 048 random rays, every method on every scenario) from the final build
O=gpurun_out/r5_c53; mkdir -p $O
timeout -k 10 1000 python tools/parity_sweep.py > $O/parity_sweep.txt 2> $O/parity_sweep.err; echo "sweep rc $?"; tail -n 6 $O/parity_sweep.txt
