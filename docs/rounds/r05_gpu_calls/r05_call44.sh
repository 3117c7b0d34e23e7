#!/bin/bash
# round 5, GPU call 44: eight compute units set aside for the re-trace by default: the whole GPU suite, the interface passes of the four methods (re-run and first-pass order), timelines
O=gpurun_out/r5_c44; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/pytest.log
{
for m in 6 2 1 8; do
python tools/bench_line.py --scenario interface --method $m --record none --steps 10 --mode plain
env RTMI_NO_DISPATCH_ORDER=1 python tools/bench_line.py --scenario interface --method $m --record none --steps 10 --mode plain
done
python tools/bench_line.py --scenario interface --record full --rec-rows 4100 --steps 5 --mode plain
python tools/bench_line.py --scenario interface --record none --steps 10
python tools/bench_line.py --scenario interface --record none --steps 10 --mode sliced
python tools/bench_line.py --scenario interface --emulate-world 8 --record none --steps 10
python tools/bench_line.py --steps 10
} 2>&1 | tee $O/times.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 8 6; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$m -o run -- python3 bench.py --scenario interface --method $m --record none --steps 2 --warmup 1 --cpu-seconds 0 --mode plain --parity-stride 0 > $O/trace$m.log 2>&1; echo "trace rc $?"
python3 tools/retrace_timeline.py $O/trace$m > $O/timeline$m.txt; tail -8 $O/timeline$m.txt
done
