#!/bin/bash
# A/B of two source trees in ONE session (each with its own librtmi.so and bench.py): tools/ab_trees.sh "<tree dirs>" <bench.py args...>
trees="$1"; shift
for t in $trees; do
  echo -n "$t [$*] : "
  (cd $t && python3 bench.py "$@" --cpu-seconds 0 2>/dev/null) | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e'%j['value'], '%.3f ms'%j['ms_per_step'], 'kern %.3f'%j['roofline']['kernel_ms_per_pass'], 'vgpr',j['roofline']['vgprs'], j['config']['launch_mode_used'], 'parity', j['parity_check']['ok'])"
done
