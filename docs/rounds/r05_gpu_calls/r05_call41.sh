#!/bin/bash
# round 5, GPU call 41: a re-run batch starts with the bundles that held its critical rays (Retrace::rot): interface passes of the four methods; the retrace tests; op9 untouched?
O=gpurun_out/r5_c41; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "critical or retrace or wave_mates or rerun" > $O/tests.txt 2>&1; echo "tests rc $?"; tail -n 3 $O/tests.txt
{
for m in 6 2 1 8; do
python tools/bench_line.py --scenario interface --method $m --record none --steps 10 --mode plain
env RTMI_NO_DISPATCH_ORDER=1 python tools/bench_line.py --scenario interface --method $m --record none --steps 10 --mode plain
done
python tools/bench_line.py --scenario interface --record full --rec-rows 4100 --steps 5 --mode plain
python tools/bench_line.py --scenario interface --record none --steps 10
python tools/bench_line.py --scenario interface --method 9 --rays 524288 --record none --steps 3 --mode plain
python tools/bench_line.py --scenario anisotropy --record none --steps 3 --mode sliced
} 2>&1 | tee $O/times.txt
env RTMI_DEBUG=1 timeout -k 10 300 python bench.py --scenario interface --method 8 --record none --steps 3 --cpu-seconds 0 --mode plain 2>&1 >/dev/null | grep "rtmi: retrace" | tail -8
