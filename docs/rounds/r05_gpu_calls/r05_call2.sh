#!/bin/bash
# round 5, GPU call 2: critical rays with the recalibrated criterion; what the re-trace costs the 1 M-ray interface fan
set -o pipefail
O=gpurun_out/r5_c2; mkdir -p $O
timeout -k 10 900 python tools/critical_ray_window.py > $O/critical.txt 2> $O/critical.err; echo "critical rc $?"
cat $O/critical.txt
run() { name=$1; shift; timeout -k 10 400 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc $?"; grep "rtmi: retrace" $O/$name.err | tail -4; }
for m in 6 1; do
run iface_op${m}_none            env RTMI_DEBUG=1 python bench.py --scenario interface --method $m --record none --steps 5 --cpu-seconds 0 --mode plain
run iface_op${m}_none_noretrace  env RTMI_NO_RETRACE=1 python bench.py --scenario interface --method $m --record none --steps 5 --cpu-seconds 0 --mode plain
done
run iface_none_auto          env RTMI_DEBUG=1 python bench.py --scenario interface --record none --steps 5 --cpu-seconds 0
run iface_full               env RTMI_DEBUG=1 python bench.py --scenario interface --record full --rec-rows 4100 --steps 5 --cpu-seconds 0 --mode plain
run iface_full_noretrace     env RTMI_NO_RETRACE=1 python bench.py --scenario interface --record full --rec-rows 4100 --steps 5 --cpu-seconds 0 --mode plain
run iface_none_sliced        env RTMI_DEBUG=1 python bench.py --scenario interface --record none --steps 5 --cpu-seconds 0 --mode sliced
run strong8_iface            env RTMI_DEBUG=1 python bench.py --scenario interface --record none --steps 5 --cpu-seconds 0 --emulate-world 8
python tools/json_brief.py $O/*.json
