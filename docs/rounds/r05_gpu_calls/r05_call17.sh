#!/bin/bash
# round 5, GPU call 17: the constant-medium step of the fused kernels (bits, then time); interface x op9's schedule
O=gpurun_out/r5_c17; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "constant_medium or critical or retrace or sharding or sliced or checkpoint or interface or lds_tile" 2>&1 | tail -4
{
tools/ab_libs.sh raytracing_amd/librtmi.so --scenario interface --record none --steps 10
RTMI_NO_RETRACE=1 tools/ab_libs.sh raytracing_amd/librtmi.so --scenario interface --record none --steps 10
tools/ab_libs.sh raytracing_amd/librtmi.so --scenario interface --method 1 --record none --steps 10
tools/ab_libs.sh raytracing_amd/librtmi.so --scenario interface --record full --rec-rows 4100 --steps 5
tools/ab_libs.sh raytracing_amd/librtmi.so --scenario interface --emulate-world 8 --record none --steps 10
tools/ab_libs.sh raytracing_amd/librtmi.so --record none --steps 10
tools/ab_libs.sh raytracing_amd/librtmi.so --steps 10
} 2>&1 | tee $O/ab.txt
for mode in auto plain sliced; do
timeout -k 10 300 python bench.py --scenario interface --method 9 --rays 524288 --record none --steps 3 --cpu-seconds 0 --mode $mode > $O/iface_op9_$mode.json 2>/dev/null
done
python tools/json_brief.py $O/*.json
RTMI_DEBUG=1 timeout -k 10 300 python bench.py --scenario interface --record none --steps 2 --cpu-seconds 0 --mode plain 2>&1 >/dev/null | grep "rtmi: retrace" | tail -5
