#!/bin/bash
# round 5, GPU call 35: op8 and op1 interface passes: timelines
O=gpurun_out/r5_c35; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 8 1; do
env RTMI_DEBUG=1 timeout -k 10 300 python bench.py --scenario interface --method $m --record none --steps 2 --cpu-seconds 0 --mode plain 2>&1 >/dev/null | grep "rtmi: retrace" | tail -7
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$m -o run -- python3 bench.py --scenario interface --method $m --record none --steps 2 --warmup 1 --cpu-seconds 0 --mode plain --parity-stride 0 > $O/trace$m.log 2>&1; echo "trace rc $?"
python3 tools/retrace_timeline.py $O/trace$m > $O/timeline$m.txt; tail -10 $O/timeline$m.txt
done
