#!/bin/bash
# round 4, GPU call 48: what do the waves of fisheye x op9 wait for with the window on?  SQC (scalar data / instruction cache) counters, window forced on and off
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4_c48; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --list-avail > $out/avail.txt 2>&1; grep -o "SQC_[A-Z_0-9]*" $out/avail.txt | sort -u | tr '\n' ' ' > $out/sqc_names.txt; cat $out/sqc_names.txt; echo
for tag in on off vert; do
  case $tag in on) a="--scenario fisheye --method 9 --rays 524288 --record none --field-path window";; off) a="--scenario fisheye --method 9 --rays 524288 --record none --field-path global";; vert) a="--method 9 --rays 524288 --record none";; esac
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $out/pmc_$tag -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 $a > $out/pmc_$tag.log 2>&1
  python3 - $out/pmc_$tag/run_counter_collection.csv $tag <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_advance' in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print(sys.argv[2], {k: f"{v / n[k]:.4g}" for k, v in acc.items()})
PY
done
