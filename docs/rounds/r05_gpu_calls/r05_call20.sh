#!/bin/bash
# round 5, GPU call 20: the same A/B with the trees in the other order (is the first run of a session the slow one?)
O=gpurun_out/r5_c20; mkdir -p $O
T="build/r04tree . build/r04tree ."
{
tools/ab_trees.sh "$T" --scenario interface --method 9 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --method 9 --rays 524288 --record none --steps 3 --mode plain
RTMI_NO_RETRACE=1 tools/ab_trees.sh "$T" --scenario interface --record none --steps 10 --mode plain
tools/ab_trees.sh "$T" --steps 10 --mode sliced
} 2>&1 | tee $O/ab.txt
