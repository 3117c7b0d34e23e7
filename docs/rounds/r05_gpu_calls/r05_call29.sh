#!/bin/bash
# round 5, GPU call 29: the hover sum weighted by |v . g| (no angle threshold), limit 14: tilted / bent walls, the straight wall's windows, the pass times
O=gpurun_out/r5_c29; mkdir -p $O
timeout -k 10 700 python tools/tilted_interface_probe.py > $O/tilted.txt 2> $O/tilted.err
echo rc=$?; cat $O/tilted.txt
{
python tools/bench_line.py --scenario interface --record none --steps 10 --mode plain
env RTMI_NO_RETRACE=1 python tools/bench_line.py --scenario interface --record none --steps 10 --mode plain
python tools/bench_line.py --scenario interface --method 1 --record none --steps 10 --mode plain
env RTMI_HOVER_LIMIT=11 python tools/bench_line.py --scenario interface --record none --steps 10 --mode plain
env RTMI_HOVER_LIMIT=17 python tools/bench_line.py --scenario interface --record none --steps 10 --mode plain
python tools/bench_line.py --steps 10
} 2>&1 | tee $O/times.txt
timeout -k 10 900 python tools/critical_ray_window.py > $O/window.txt 2> $O/window.err
echo rc=$?; cat $O/window.txt
