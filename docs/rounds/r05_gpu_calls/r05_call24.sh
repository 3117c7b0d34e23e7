#!/bin/bash
# round 5, GPU call 24: the whole GPU suite on the present build; interface x op9 once more against round 4's tree
O=gpurun_out/r5_c24; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/pytest.log
T="build/r04tree ."
{
tools/ab_trees.sh "$T" --scenario interface --method 9 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --method 9 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --scenario interface --method 5 --rays 524288 --record none --steps 3 --mode plain
} 2>&1 | tee $O/ab.txt
