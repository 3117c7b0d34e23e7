#!/bin/bash
# round 5, GPU call 3: the reference-order step's flat path (bits), critical rays, the re-trace beside the main kernel
set -o pipefail
O=gpurun_out/r5_c3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_exact.py -x -q > $O/pytest_exact.log 2>&1; echo "pytest exact rc $?"; tail -3 $O/pytest_exact.log
timeout -k 10 900 python tools/critical_ray_window.py > $O/critical.txt 2> $O/critical.err; echo "critical rc $?"
grep -v "^#      ray" $O/critical.txt
run() { name=$1; shift; timeout -k 10 400 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc $?"; grep "rtmi: retrace" $O/$name.err | tail -5; }
for m in 6 1; do
run iface_op${m}_none            env RTMI_DEBUG=1 python bench.py --scenario interface --method $m --record none --steps 5 --cpu-seconds 0 --mode plain
run iface_op${m}_none_noretrace  env RTMI_NO_RETRACE=1 python bench.py --scenario interface --method $m --record none --steps 5 --cpu-seconds 0 --mode plain
run iface_op${m}_none_reforder   python bench.py --scenario interface --method $m --record none --steps 3 --cpu-seconds 0 --reference-order
done
run iface_full               env RTMI_DEBUG=1 python bench.py --scenario interface --record full --rec-rows 4100 --steps 5 --cpu-seconds 0 --mode plain
run iface_full_noretrace     env RTMI_NO_RETRACE=1 python bench.py --scenario interface --record full --rec-rows 4100 --steps 5 --cpu-seconds 0 --mode plain
run strong8_iface            env RTMI_DEBUG=1 python bench.py --scenario interface --record none --steps 5 --cpu-seconds 0 --emulate-world 8
run vert_op6_reforder        python bench.py --record none --steps 3 --cpu-seconds 0 --reference-order
python tools/json_brief.py $O/*.json
