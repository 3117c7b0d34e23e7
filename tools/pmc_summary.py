#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel, per-counter averages over dispatches."""
import csv, glob, sys, collections
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc*/runc/*_counter_collection.csv"
agg = collections.defaultdict(list)
for fn in sorted(glob.glob(pat)):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    if "k_advance" in k or (len(sys.argv) > 2 and sys.argv[2] == "all"):
        print(f"{k:42s} {c:24s} n={len(v):3d} avg={sum(v)/len(v):.6g} last={v[-1]:.6g}")
