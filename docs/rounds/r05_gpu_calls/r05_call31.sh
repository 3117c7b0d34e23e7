#!/bin/bash
# round 5, GPU call 31: the count published in order (push_critical under its lock), limit 12: pass time against the limit, the retrace tests, tilted walls
O=gpurun_out/r5_c31; mkdir -p $O
{
for lim in 12 10 14 17; do
echo "limit $lim"; env RTMI_HOVER_LIMIT=$lim python tools/bench_line.py --scenario interface --record none --steps 10 --mode plain
done
python tools/bench_line.py --scenario interface --method 1 --record none --steps 10 --mode plain
env RTMI_NO_RETRACE=1 python tools/bench_line.py --scenario interface --record none --steps 10 --mode plain
} 2>&1 | tee $O/times.txt
env RTMI_DEBUG=1 timeout -k 10 300 python bench.py --scenario interface --record none --steps 3 --cpu-seconds 0 --mode plain 2>&1 >/dev/null | grep "rtmi: retrace" | tail -10
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "critical or retrace or wave_mates" 2>&1 | tail -n 5
timeout -k 10 700 python tools/tilted_interface_probe.py > $O/tilted.txt 2> $O/tilted.err
echo rc=$?; cat $O/tilted.txt
