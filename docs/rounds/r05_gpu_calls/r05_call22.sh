#!/bin/bash
# round 5, GPU call 22: which of this round's commits slowed interface x op9 down (one tree per commit under build/, each with its own library)
O=gpurun_out/r5_c22; mkdir -p $O
T="build/r04tree build/bis_8a19b8e build/bis_89c3066 build/bis_f19b067 build/bis_21d0a09 ."
{
tools/ab_trees.sh "$T" --scenario interface --method 9 --rays 524288 --record none --steps 3 --mode plain
RTMI_NO_RETRACE=1 tools/ab_trees.sh "$T" --scenario interface --record none --steps 10 --mode plain
} 2>&1 | tee $O/bisect.txt
