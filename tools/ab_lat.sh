for cfg in "--rays 65536 --record none" "--rays 65536" "--total-rays 1048576 --emulate-world 8" "--total-rays 1048576 --emulate-world 8 --record none" "--total-rays 1048576 --emulate-world 4 --record none" "--rays 32768 --record none"; do
  for lat in 0 1; do
    if [ $lat = 0 ]; then export RTMI_NO_LAT=1; else unset RTMI_NO_LAT; fi
    echo -n "lat=$lat [$cfg] : "
    python3 bench.py $cfg --steps 10 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e'%j['value'], '%.3f ms'%j['ms_per_step'], 'kern %.3f'%j['roofline']['kernel_ms_per_pass'], 'vgpr',j['roofline']['vgprs'], j['config']['launch_mode_used'], 'parity', j['parity_check']['ok'])"
  done
done
