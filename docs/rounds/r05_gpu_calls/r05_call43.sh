#!/bin/bash
# round 5, GPU call 43: compute units set aside for the re-trace (RTMI_RETRACE_CUS = n: CU masks on the aux streams and on the main kernel's): op8 / op6 / op1
O=gpurun_out/r5_c43; mkdir -p $O
{
for m in 8 6; do
for n in 0 1 2 4 8 16; do
echo -n "cus $n: "; env RTMI_RETRACE_CUS=$n timeout -k 10 200 python tools/bench_line.py --scenario interface --method $m --record none --steps 10 --mode plain
done
done
echo -n "cus 4: "; env RTMI_RETRACE_CUS=4 timeout -k 10 200 python tools/bench_line.py --scenario interface --method 1 --record none --steps 10 --mode plain
} 2>&1 | tee $O/times.txt
env RTMI_RETRACE_CUS=4 RTMI_DEBUG=1 timeout -k 10 300 python bench.py --scenario interface --method 8 --record none --steps 3 --cpu-seconds 0 --mode plain 2>&1 >/dev/null | grep "rtmi: retrace" | tail -6
env RTMI_RETRACE_CUS=4 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "critical or retrace or wave_mates or rerun" > $O/tests.txt 2>&1; echo "tests rc $?"; tail -n 3 $O/tests.txt
