#!/bin/bash
# call 28: the hand-over on tilted and bent walls (fields from samples)
mkdir -p gpurun_out/r5_c28
timeout -k 10 900 python tools/tilted_interface_probe.py > gpurun_out/r5_c28/tilted.txt 2> gpurun_out/r5_c28/tilted.err
echo rc=$?
tail -30 gpurun_out/r5_c28/tilted.txt
tail -5 gpurun_out/r5_c28/tilted.err
