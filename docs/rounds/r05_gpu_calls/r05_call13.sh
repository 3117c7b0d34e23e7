#!/bin/bash
# round 5, GPU call 13: the few-waves kernel without the map's tests for fields that have none; the critical check off the hot path
O=gpurun_out/r5_c13; mkdir -p $O
L="raytracing_amd/librtmi.so"
{
tools/ab_libs.sh "$L" --total-rays 1048576 --emulate-world 8 --steps 20
tools/ab_libs.sh "$L" --total-rays 1048576 --emulate-world 8 --record none --steps 20
tools/ab_libs.sh "$L" --rays 65536 --steps 20
tools/ab_libs.sh "$L" --rays 65536 --record none --steps 20
tools/ab_libs.sh "$L" --scenario interface --total-rays 1048576 --emulate-world 8 --record none --steps 10
tools/ab_libs.sh "$L" --scenario interface --record none --steps 10
RTMI_NO_RETRACE=1 tools/ab_libs.sh "$L" --scenario interface --record none --steps 10
tools/ab_libs.sh "$L" --scenario interface --method 1 --record none --steps 10
tools/ab_libs.sh "$L" --scenario interface --record full --rec-rows 4100 --steps 5
tools/ab_libs.sh "$L" --scenario fisheye --record none --steps 10
tools/ab_libs.sh "$L" --record none --steps 10
} 2>&1 | tee $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "critical or retrace or sharding or checkpoint" 2>&1 | tail -3
