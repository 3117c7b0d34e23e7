#!/bin/bash
# round 5, GPU call 11: the whole GPU suite on the build with the re-trace, then reference-order timings (the NOFLAT builds), the default line
O=gpurun_out/r5_c11; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
run() { name=$1; shift; timeout -k 10 400 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc $?"; }
run vert_op6_reforder        python bench.py --record none --steps 3 --cpu-seconds 0 --reference-order
run iface_op6_reforder       python bench.py --scenario interface --record none --steps 3 --cpu-seconds 0 --reference-order
run vert_op7_default         python bench.py --method 7 --record none --steps 3 --cpu-seconds 0
run iface_op6_none           python bench.py --scenario interface --record none --steps 5 --cpu-seconds 0
run bench_default            python bench.py
python tools/json_brief.py $O/*.json
