#!/bin/bash
# round 5, GPU call 38: the round's profile set again, from the final build (kernel trace + PMC passes per configuration); the 1 M parity sweep
O=gpurun_out/r5_c38; mkdir -p $O
tools/r05_profile_all.sh > $O/profile_all.log 2>&1; echo "profiles rc $?"; tail -n 3 $O/profile_all.log
timeout -k 10 600 python tools/parity_sweep_1m.py > $O/parity_sweep_1m.txt 2> $O/parity_sweep_1m.err; echo "sweep rc $?"; tail -n 12 $O/parity_sweep_1m.txt
