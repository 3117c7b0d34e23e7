#!/bin/bash
# round 5, GPU call 15: parity sweeps at full size (kept under profiles/), the default line, a thread-trace attempt on the few-waves kernel
O=gpurun_out/r5_c15; mkdir -p $O
timeout -k 10 600 python tools/parity_sweep_1m.py > $O/parity_sweep_1m.txt 2> $O/parity_sweep_1m.err; echo "sweep1m rc $?"; tail -3 $O/parity_sweep_1m.txt | cut -c1-250
timeout -k 10 400 python tools/parity_sweep.py > $O/parity_sweep.txt 2> $O/parity_sweep.err; echo "sweep rc $?"; tail -2 $O/parity_sweep.txt | cut -c1-250
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python tools/json_brief.py $O/bench_default.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 240 rocprofv3 --att --att-target-cu 1 --kernel-include-regex "k_advance_lat" -d $O/att -o run -- python3 bench.py --rays 65536 --steps 1 --warmup 0 --cpu-seconds 0 --parity-stride 0 --mode plain > $O/att.log 2>&1; echo "att rc $?"
tail -5 $O/att.log; find $O/att -type f | head -20; du -sh $O/att
