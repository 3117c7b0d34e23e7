#!/bin/bash
# round 5, GPU call 33: the full-record interface pass with the per-lane kept cell in the tails: timeline
O=gpurun_out/r5_c33; mkdir -p $O
{
python tools/bench_line.py --scenario interface --record full --rec-rows 4100 --steps 5 --mode plain
env RTMI_NO_RETRACE=1 python tools/bench_line.py --scenario interface --record full --rec-rows 4100 --steps 5 --mode plain
python tools/bench_line.py --scenario interface --record stride:16 --steps 5 --mode plain
} 2>&1 | tee $O/times.txt
env RTMI_DEBUG=1 timeout -k 10 300 python bench.py --scenario interface --record full --rec-rows 4100 --steps 2 --cpu-seconds 0 --mode plain 2>&1 >/dev/null | grep "rtmi: retrace" | tail -8
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 bench.py --scenario interface --record full --rec-rows 4100 --steps 2 --warmup 1 --cpu-seconds 0 --mode plain --parity-stride 0 > $O/trace.log 2>&1; echo "trace rc $?"
python3 tools/retrace_timeline.py $O/trace > $O/timeline.txt; tail -12 $O/timeline.txt
