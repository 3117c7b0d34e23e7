#!/bin/bash
# A/B of library builds in ONE session: tools/ab_libs.sh "<lib paths, space separated>" <bench.py args...>   (RTMI_LIB_PATH selects the build)
libs="$1"; shift
for lib in $libs; do
  echo -n "$(basename $lib) [$*] : "
  RTMI_LIB_PATH=$lib python3 bench.py "$@" --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e'%j['value'], '%.3f ms'%j['ms_per_step'], 'kern %.3f'%j['roofline']['kernel_ms_per_pass'], 'vgpr',j['roofline']['vgprs'], j['config']['launch_mode_used'], 'parity', j['parity_check']['ok'])"
done
