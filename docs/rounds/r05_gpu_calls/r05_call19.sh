#!/bin/bash
# round 5, GPU call 19: interface x op9 in this tree and in round 4's: times again, then instruction and wait counters of both
O=gpurun_out/r5_c19; mkdir -p $O
T=". build/r04tree"
{
tools/ab_trees.sh "$T" --scenario interface --method 9 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --scenario interface --method 5 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --method 9 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --scenario interface --record none --steps 10 --mode plain
} 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in . build/r04tree; do
  tag=$(basename $(realpath $t))
  (cd $t && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$tag -o run -- python3 bench.py --scenario interface --method 9 --rays 524288 --record none --steps 2 --warmup 1 --cpu-seconds 0 --parity-stride 0 --mode plain > $GRAFT_REPO_ROOT/$O/pmc_$tag.log 2>&1); echo "pmc $tag rc $?"
done
python3 - <<'PY'
import csv, glob, collections
for tag in ("repo", "r04tree", "neyuru__RayTracing"):
    agg = collections.defaultdict(list)
    for fn in glob.glob(f"gpurun_out/r5_c19/pmc_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "k_advance" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if agg: print(tag, {k: f"{sum(v)/len(v):.4g}" for k, v in sorted(agg.items())})
PY
ls $O
