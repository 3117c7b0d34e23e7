#!/bin/bash
# round 5, GPU call 1: the suite on the new build, the critical-ray window with the automatic re-trace, its cost, the default line
set -o pipefail
mkdir -p gpurun_out/r5_c1
O=gpurun_out/r5_c1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
RTMI_DEBUG=1 timeout -k 10 600 python tools/critical_ray_window.py > $O/critical.txt 2> $O/critical.err; echo "critical rc $?"
cat $O/critical.txt
for nr in 0 1; do
  for rec in none full; do
    $( [ $nr = 1 ] && echo env RTMI_NO_RETRACE=1 ) timeout -k 10 300 python bench.py --scenario interface --record $rec --steps 5 --cpu-seconds 0 --mode plain > $O/iface_${rec}_nr$nr.json 2> $O/iface_${rec}_nr$nr.err; echo "iface $rec noretrace=$nr rc $?"
  done
done
timeout -k 10 300 python bench.py --scenario interface --record none --steps 5 --cpu-seconds 0 > $O/iface_none_auto.json 2> $O/iface_none_auto.err; echo "iface auto rc $?"
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc $?"
python tools/json_brief.py $O/*.json
