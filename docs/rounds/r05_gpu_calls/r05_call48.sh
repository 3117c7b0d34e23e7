#!/bin/bash
# round 5, GPU call 48: the tails' row counter counted down instead of divided: first-pass and re-run interface times, the retrace tests
O=gpurun_out/r5_c48; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "critical or retrace or wave_mates or rerun" > $O/tests.txt 2>&1; echo "tests rc $?"; tail -n 3 $O/tests.txt
{
for m in 6 8; do
python tools/bench_line.py --scenario interface --method $m --record none --steps 10 --mode plain
env RTMI_NO_DISPATCH_ORDER=1 python tools/bench_line.py --scenario interface --method $m --record none --steps 10 --mode plain
done
env RTMI_NO_DISPATCH_ORDER=1 python tools/bench_line.py --scenario interface --record full --rec-rows 4100 --steps 5 --mode plain
} 2>&1 | tee $O/times.txt
env RTMI_NO_DISPATCH_ORDER=1 RTMI_DEBUG=1 timeout -k 10 300 python bench.py --scenario interface --record none --steps 2 --cpu-seconds 0 --mode plain 2>&1 >/dev/null | grep "rtmi: retrace" | tail -5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
env RTMI_NO_DISPATCH_ORDER=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace6 -o run -- python3 bench.py --scenario interface --record none --steps 2 --warmup 1 --cpu-seconds 0 --mode plain --parity-stride 0 > $O/trace6.log 2>&1; echo "trace rc $?"
python3 tools/retrace_timeline.py $O/trace6 > $O/timeline6.txt; tail -9 $O/timeline6.txt
