#!/bin/bash
# round 5, GPU call 30: the weighted hover sum's pass time against its limit (twice each), and the timeline of a pass at 14
O=gpurun_out/r5_c30; mkdir -p $O
{
for rep in 1 2; do
for lim in 14 11 17 20; do
echo "limit $lim"; env RTMI_HOVER_LIMIT=$lim python tools/bench_line.py --scenario interface --record none --steps 10 --mode plain
done
done
} 2>&1 | tee $O/times.txt
env RTMI_DEBUG=1 timeout -k 10 300 python bench.py --scenario interface --record none --steps 3 --cpu-seconds 0 --mode plain 2>&1 >/dev/null | grep "rtmi: retrace" | tail -24
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 bench.py --scenario interface --record none --steps 2 --warmup 1 --cpu-seconds 0 --mode plain --parity-stride 0 > $O/trace.log 2>&1; echo "trace rc $?"
python3 tools/retrace_timeline.py $O/trace > $O/timeline.txt; tail -40 $O/timeline.txt
