#!/bin/bash
# round 5, GPU call 21: interface x op9, this tree vs round 4's: kernel-trace durations, instruction-cache and busy counters
O=gpurun_out/r5_c21; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A="--scenario interface --method 9 --rays 524288 --record none --steps 2 --warmup 1 --cpu-seconds 0 --parity-stride 0 --mode plain"
for t in . build/r04tree; do
  tag=$(basename $(realpath $t))
  (cd $t && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_$tag -o run -- python3 bench.py $A > $GRAFT_REPO_ROOT/$O/kt_$tag.log 2>&1)
  (cd $t && rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_IFETCH SQ_WAIT_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$tag -o run -- python3 bench.py $A > $GRAFT_REPO_ROOT/$O/pmc_$tag.log 2>&1); echo "$tag rc $?"
done
python3 - <<'PY'
import csv, glob, collections
for tag in ("repo", "neyuru__RayTracing", "r04tree"):
    for fn in glob.glob(f"gpurun_out/r5_c21/kt_{tag}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_advance" in r["Name"]: print(tag, "kernel-trace:", r["Name"][:40], "calls", r["Calls"], "avg ns", r["AverageNs"])
    agg = collections.defaultdict(list)
    for fn in glob.glob(f"gpurun_out/r5_c21/pmc_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_advance" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if agg: print(tag, {k: f"{sum(v)/len(v):.4g}" for k, v in sorted(agg.items())})
PY
tail -3 $O/pmc_*.log | head -20
