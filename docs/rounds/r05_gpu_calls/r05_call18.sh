#!/bin/bash
# round 5, GPU call 18: this tree against round 4's (build/r04tree: git archive 327a1f4, its own library), one session
O=gpurun_out/r5_c18; mkdir -p $O
T=". build/r04tree"
{
tools/ab_trees.sh "$T" --scenario interface --method 9 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --scenario interface --record none --steps 10 --mode plain
tools/ab_trees.sh "$T" --scenario interface --method 1 --record none --steps 10 --mode plain
tools/ab_trees.sh "$T" --scenario interface --method 7 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --scenario interface --method 3 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --scenario interface --method 5 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --scenario anisotropy --record none --steps 3 --mode sliced
tools/ab_trees.sh "$T" --method 9 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --record none --steps 10 --mode plain
tools/ab_trees.sh "$T" --steps 10 --mode sliced
tools/ab_trees.sh "$T" --scenario fisheye --record none --steps 10 --mode sliced
tools/ab_trees.sh "$T" --dtype f32 --rays 8388608 --record none --steps 5 --mode plain
} 2>&1 | tee $O/ab_r04_r05.txt
