#!/bin/bash
# round 5, GPU call 32 (second run): per-lane kept cell with prefetch in the fused tails of re-traced rays, hand-back after 256 steps
O=gpurun_out/r5_c32; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "critical or retrace or wave_mates" > $O/tests.txt 2>&1; echo "tests rc $?"; tail -n 5 $O/tests.txt
{
python tools/bench_line.py --scenario interface --record none --steps 10 --mode plain
env RTMI_NO_RETRACE=1 python tools/bench_line.py --scenario interface --record none --steps 10 --mode plain
python tools/bench_line.py --scenario interface --method 1 --record none --steps 10 --mode plain
env RTMI_NO_RETRACE=1 python tools/bench_line.py --scenario interface --method 1 --record none --steps 10 --mode plain
python tools/bench_line.py --scenario interface --record full --rec-rows 4100 --steps 5 --mode plain
} 2>&1 | tee $O/times.txt
env RTMI_DEBUG=1 timeout -k 10 300 python bench.py --scenario interface --record none --steps 3 --cpu-seconds 0 --mode plain 2>&1 >/dev/null | grep "rtmi: retrace\|rtmi: field" | tail -8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 bench.py --scenario interface --record none --steps 2 --warmup 1 --cpu-seconds 0 --mode plain --parity-stride 0 > $O/trace.log 2>&1; echo "trace rc $?"
python3 tools/retrace_timeline.py $O/trace > $O/timeline.txt; tail -12 $O/timeline.txt
