#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c8
A=build/ab
{
tools/ab_variants.sh "--scenario anisotropy --record none --steps 3" $A/librtmi_base.so $A/librtmi_momB.so $A/librtmi_pred2.so
tools/ab_variants.sh "--scenario anisotropy --method 10 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_momB.so $A/librtmi_pred2.so
} > gpurun_out/r4_c8/ab.txt 2>&1
cat gpurun_out/r4_c8/ab.txt
