#!/bin/bash
# round 5, GPU call 36: the whole GPU suite on the present build, the default bench line, the critical-ray windows, tilted walls
O=gpurun_out/r5_c36; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/pytest.log
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python tools/json_brief.py $O/bench_default.json
timeout -k 10 600 python tools/critical_ray_window.py > $O/window.txt 2> $O/window.err; echo "window rc $?"; grep -v "^#      ray" $O/window.txt | tail -16
timeout -k 10 600 python tools/tilted_interface_probe.py > $O/tilted.txt 2> $O/tilted.err; echo "tilted rc $?"; cat $O/tilted.txt
