#!/bin/bash
# round 5, GPU call 14: the profile set of this round's kernels; is the thread trace available?; the scalar-cache window on fisheye x op9 (counters)
O=gpurun_out/r5_c14; mkdir -p $O
rocprofv3 --help 2>&1 | grep -i -A2 "att\b\|--att\|thread-trace\|advanced" | head -30 > $O/rocprof_att_help.txt
tools/r05_profile_all.sh > $O/profile_all.log 2>&1; echo "profiles rc $?"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for fp in window global; do
  rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_VALU SQ_WAIT_ANY --output-format csv -d $O/win_$fp -o run -- python3 bench.py --scenario fisheye --method 9 --rays 524288 --record none --steps 2 --warmup 1 --cpu-seconds 0 --parity-stride 0 --field-path $fp > $O/win_$fp.log 2>&1; echo "pmc $fp rc $?"
done
python3 - <<'PY'
import csv, glob, collections
for fp in ("window", "global"):
    agg = collections.defaultdict(list)
    for fn in glob.glob(f"gpurun_out/r5_c14/win_{fp}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_advance" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(fp, {k: f"{sum(v)/len(v):.4g}" for k, v in sorted(agg.items())})
PY
ls gpurun_out | grep prof_r05
