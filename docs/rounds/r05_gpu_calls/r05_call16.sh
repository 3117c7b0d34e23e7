#!/bin/bash
# round 5, GPU call 16: the re-trace kernels alone on the chip (kernel trace); interface x op9 and cfg5 against round 4's times
O=gpurun_out/r5_c16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python3 tools/retrace_standalone_probe.py 6 > $O/standalone.log 2>&1; echo "trace rc $?"; grep "^op" $O/standalone.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r5_c16/trace/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = None
out = []
for r in rows:
    n = r['Kernel_Name']
    if 'k_advance' in n or 'k_retrace' in n or 'k_init' in n:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if 'k_init' in n: t0 = s
        out.append(f"{(s - (t0 or s)) / 1e6:9.3f} .. {(e - (t0 or s)) / 1e6:9.3f} ms  grid {r.get('Grid_Size_X', r.get('Grid_Size'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size'))}  {n[:60]}")
open('gpurun_out/r5_c16/standalone_timeline.txt', 'w').write("\n".join(out) + "\n")
print("\n".join(out[-8:]))
PY
{
tools/ab_libs.sh raytracing_amd/librtmi.so --scenario interface --method 9 --rays 524288 --record none --steps 3
tools/ab_libs.sh raytracing_amd/librtmi.so --scenario anisotropy --record none --steps 3
tools/ab_libs.sh raytracing_amd/librtmi.so --scenario fisheye --method 9 --rays 524288 --record none --steps 3
tools/ab_libs.sh raytracing_amd/librtmi.so --method 3 --record none --steps 3
} 2>&1 | tee $O/exact_methods.txt
