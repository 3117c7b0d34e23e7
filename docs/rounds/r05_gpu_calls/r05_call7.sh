#!/bin/bash
O=gpurun_out/r5_c7; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== default build" | tee $O/probe.txt
timeout -k 10 300 python -m pytest tests/test_gpu_exact.py -x -q 2>&1 | tail -2
run() { name=$1; shift; timeout -k 10 400 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc $?"; grep "rtmi: retrace" $O/$name.err | tail -6; }
run iface_op6_none            env RTMI_DEBUG=1 python bench.py --scenario interface --method 6 --record none --steps 5 --cpu-seconds 0 --mode plain
run iface_op6_none_noretrace  env RTMI_NO_RETRACE=1 python bench.py --scenario interface --method 6 --record none --steps 5 --cpu-seconds 0 --mode plain
run iface_op1_none            env RTMI_DEBUG=1 python bench.py --scenario interface --method 1 --record none --steps 5 --cpu-seconds 0 --mode plain
run iface_op1_none_noretrace  env RTMI_NO_RETRACE=1 python bench.py --scenario interface --method 1 --record none --steps 5 --cpu-seconds 0 --mode plain
run iface_full               env RTMI_DEBUG=1 python bench.py --scenario interface --record full --rec-rows 4100 --steps 5 --cpu-seconds 0 --mode plain
run iface_full_noretrace     env RTMI_NO_RETRACE=1 python bench.py --scenario interface --record full --rec-rows 4100 --steps 5 --cpu-seconds 0 --mode plain
METHODS=6,1,2,8 timeout -k 10 900 python tools/critical_ray_window.py > $O/critical.txt 2> $O/critical.err; echo "critical rc $?"
grep -v "^#      ray" $O/critical.txt | tail -14
python tools/json_brief.py $O/*.json
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 bench.py --scenario interface --method 6 --record none --steps 2 --warmup 1 --cpu-seconds 0 --mode plain --parity-stride 0 > $O/trace.log 2>&1; echo "trace rc $?"
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r5_c7/trace/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = None
out = []
for r in rows:
    n = r['Kernel_Name']
    if 'k_advance' in n or 'k_retrace' in n or 'k_init' in n:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if 'k_init' in n: t0 = s
        out.append(f"{(s - (t0 or s)) / 1e6:9.3f} .. {(e - (t0 or s)) / 1e6:9.3f} ms  grid {r.get('Grid_Size_X', r.get('Grid_Size'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size'))} queue {r.get('Queue_Id')}  {n[:70]}")
open('gpurun_out/r5_c7/trace_timeline.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out[-40:]))
PY
