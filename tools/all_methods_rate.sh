#!/bin/bash
# ray-steps/s of every step method on the scenarios it applies to, 1 048 576 rays (524 288 for the golden-section methods), fp64,
# no record, default schedule: one line per case through tools/bench_line.py
for scen in vert_heterogeneous fisheye interface; do
  for m in 1 2 3 4 5 6 7 8 9; do
    rays=1048576; case $m in 5|9) rays=524288;; esac
    echo -n "$scen : "; python3 tools/bench_line.py --scenario $scen --method $m --rays $rays --record none --steps 3
  done
done
for m in 10 11; do echo -n "anisotropy : "; python3 tools/bench_line.py --scenario anisotropy --method $m --rays 524288 --record none --steps 3; done
# rtmi_params.reference_order = 1: op1/2/6/8 in the reference's operation order too (op7's default is that already)
for m in 1 2 6 8; do echo -n "vert_heterogeneous reference_order : "; python3 tools/bench_line.py --scenario vert_heterogeneous --method $m --rays 1048576 --record none --steps 3 --reference-order; done
# op7's opt-ins: the fused form (--fused) and the reference-order step on the fused field lookup (--fast-field)
for scen in vert_heterogeneous fisheye interface; do
  for o in --fused --fast-field; do echo -n "$scen op7 $o : "; python3 tools/bench_line.py --scenario $scen --method 7 --rays 1048576 --record none --steps 3 $o; done
done
