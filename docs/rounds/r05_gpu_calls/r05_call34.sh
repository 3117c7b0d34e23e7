#!/bin/bash
# round 5, GPU call 34: chunks only onto idle streams: the full-record and no-record interface passes, the retrace tests
O=gpurun_out/r5_c34; mkdir -p $O
{
python tools/bench_line.py --scenario interface --record full --rec-rows 4100 --steps 5 --mode plain
python tools/bench_line.py --scenario interface --record none --steps 10 --mode plain
python tools/bench_line.py --scenario interface --method 1 --record none --steps 10 --mode plain
python tools/bench_line.py --scenario interface --method 2 --record none --steps 10 --mode plain
env RTMI_NO_RETRACE=1 python tools/bench_line.py --scenario interface --method 2 --record none --steps 10 --mode plain
python tools/bench_line.py --scenario interface --method 8 --record none --steps 10 --mode plain
env RTMI_NO_RETRACE=1 python tools/bench_line.py --scenario interface --method 8 --record none --steps 10 --mode plain
python tools/bench_line.py --scenario interface --record stride:16 --steps 5
} 2>&1 | tee $O/times.txt
env RTMI_DEBUG=1 timeout -k 10 300 python bench.py --scenario interface --record full --rec-rows 4100 --steps 2 --cpu-seconds 0 --mode plain 2>&1 >/dev/null | grep "rtmi: retrace" | tail -8
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "critical or retrace or wave_mates" > $O/tests.txt 2>&1; echo "tests rc $?"; tail -n 3 $O/tests.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 bench.py --scenario interface --record full --rec-rows 4100 --steps 2 --warmup 1 --cpu-seconds 0 --mode plain --parity-stride 0 > $O/trace.log 2>&1; echo "trace rc $?"
python3 tools/retrace_timeline.py $O/trace > $O/timeline.txt; tail -10 $O/timeline.txt
