#!/bin/bash
# round 5, GPU call 27 (the flat path flags in Ray::rn, no new Ray members): FieldDev and Consts back at round 4 layouts (gflat and the hover limit travel with the gather): interface x op9 against round 4's tree; the bit tests
O=gpurun_out/r5_c27; mkdir -p $O
T="build/r04tree ."
{
tools/ab_trees.sh "$T" --scenario interface --method 9 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --method 9 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --scenario interface --method 5 --rays 524288 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --scenario anisotropy --record none --steps 3 --mode sliced
tools/ab_trees.sh "$T" --scenario interface --method 3 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --method 7 --record none --steps 3 --mode plain
tools/ab_trees.sh "$T" --scenario interface --record none --steps 10 --mode plain --reference-order
tools/ab_trees.sh "$T" --scenario interface --record none --steps 10 --mode plain
} 2>&1 | tee $O/ab.txt
timeout -k 10 600 python -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -x -q -k "exact or critical or retrace or oracle_bits or reference" 2>&1 | tail -n 3
