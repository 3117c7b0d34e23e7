#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv directory: the advance / re-trace kernels of each pass on one time axis (ms from the pass's k_init)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    n = r["Kernel_Name"]
    if "k_advance" in n or "k_retrace" in n or "k_init" in n:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "k_init" in n:
            t0 = s
        print(f"{(s - (t0 or s)) / 1e6:9.3f} .. {(e - (t0 or s)) / 1e6:9.3f} ms  grid {r.get('Grid_Size_X', r.get('Grid_Size'))} wg "
              f"{r.get('Workgroup_Size_X', r.get('Workgroup_Size'))} queue {r.get('Queue_Id')}  {n[:60]}")
