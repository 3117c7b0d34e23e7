#!/bin/bash
# round 5, GPU call 50: the final build: the whole GPU suite, smoke(), the default bench line, the interface line
O=gpurun_out/r5_c50; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python tools/json_brief.py $O/bench_default.json
timeout -k 10 300 python bench.py --scenario interface --record none --cpu-seconds 0 > $O/bench_iface.json 2> $O/bench_iface.err; echo "bench rc $?"; python tools/json_brief.py $O/bench_iface.json
timeout -k 10 300 python bench.py --emulate-world 8 --cpu-seconds 0 > $O/bench_strong8.json 2> $O/bench_strong8.err; echo "bench rc $?"; python tools/json_brief.py $O/bench_strong8.json
