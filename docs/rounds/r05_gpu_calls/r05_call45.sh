#!/bin/bash
# round 5, GPU call 45: the interface profiles, the critical-ray windows, tilted walls, the 1 M parity sweep and the default bench line from the final build
O=gpurun_out/r5_c45; mkdir -p $O
tools/profile_config.sh r05_iface_none --scenario interface --record none > $O/prof_none.log 2>&1; echo "prof none rc $?"
tools/profile_config.sh r05_iface_full --scenario interface --record full --rec-rows 4100 > $O/prof_full.log 2>&1; echo "prof full rc $?"
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python tools/json_brief.py $O/bench_default.json
timeout -k 10 600 python tools/critical_ray_window.py > $O/window.txt 2> $O/window.err; echo "window rc $?"; grep -v "^#      ray" $O/window.txt | tail -16
timeout -k 10 600 python tools/tilted_interface_probe.py > $O/tilted.txt 2> $O/tilted.err; echo "tilted rc $?"; tail -n 17 $O/tilted.txt
timeout -k 10 600 python tools/parity_sweep_1m.py > $O/parity_sweep_1m.txt 2> $O/parity_sweep_1m.err; echo "sweep rc $?"; tail -n 3 $O/parity_sweep_1m.txt
