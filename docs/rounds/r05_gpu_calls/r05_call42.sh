#!/bin/bash
# round 5, GPU call 42: op8 with the learnt dispatch order: timeline
O=gpurun_out/r5_c42; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 8 6; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$m -o run -- python3 bench.py --scenario interface --method $m --record none --steps 3 --warmup 1 --cpu-seconds 0 --mode plain --parity-stride 0 > $O/trace$m.log 2>&1; echo "trace rc $?"
python3 tools/retrace_timeline.py $O/trace$m > $O/timeline$m.txt; tail -16 $O/timeline$m.txt
done
