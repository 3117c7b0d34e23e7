#!/bin/bash
# round 5, GPU call 23: FieldDev padded back to a multiple of 64 bytes: interface x op9 and the other register-bound kernels against round 4's tree
O=gpurun_out/r5_c23; mkdir -p $O
T="build/r04tree build/bis_21d0a09 ."
{
tools/ab_trees.sh "$T" --scenario interface --method 9 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --method 9 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --scenario interface --method 5 --rays 524288 --record none --steps 3 --mode plain
RTMI_NO_RETRACE=1 tools/ab_trees.sh "$T" --scenario interface --record none --steps 10 --mode plain
tools/ab_trees.sh "$T" --scenario anisotropy --record none --steps 3 --mode sliced
tools/ab_trees.sh "$T" --steps 10 --mode sliced
tools/ab_trees.sh "$T" --record none --steps 10 --mode plain
tools/ab_trees.sh "$T" --rays 65536 --steps 20
tools/ab_trees.sh "$T" --scenario fisheye --record none --steps 10 --mode sliced
} 2>&1 | tee $O/ab.txt
