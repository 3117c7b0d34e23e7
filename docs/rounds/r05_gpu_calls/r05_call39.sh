#!/bin/bash
# round 5, GPU call 39: the hover sum taken at the end of the step, per-slot flags instead of the published count: op1 / op8 out of their spills?
O=gpurun_out/r5_c39; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "critical or retrace or wave_mates" > $O/tests.txt 2>&1; echo "tests rc $?"; tail -n 3 $O/tests.txt
{
for m in 6 2 1 8; do
python tools/bench_line.py --scenario interface --method $m --record none --steps 10 --mode plain
env RTMI_NO_RETRACE=1 python tools/bench_line.py --scenario interface --method $m --record none --steps 10 --mode plain
done
python tools/bench_line.py --scenario interface --record full --rec-rows 4100 --steps 5 --mode plain
python tools/bench_line.py --scenario interface --emulate-world 8 --record none --steps 10
} 2>&1 | tee $O/times.txt
env RTMI_DEBUG=1 timeout -k 10 300 python bench.py --scenario interface --method 8 --record none --steps 2 --cpu-seconds 0 --mode plain 2>&1 >/dev/null | grep "rtmi: retrace" | tail -7
timeout -k 10 600 python tools/critical_ray_window.py > $O/window.txt 2> $O/window.err; echo "window rc $?"; grep -v "^#      ray" $O/window.txt | tail -16
