#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c10
A=build/ab
{
echo "## inl: glibc's sin/cos inline (range-specialised) in the reference-order step instead of a call"
tools/ab_variants.sh "--method 7 --record none --steps 3" $A/librtmi_base.so $A/librtmi_inl.so
tools/ab_variants.sh "--method 3 --record none --steps 3" $A/librtmi_base.so $A/librtmi_inl.so
tools/ab_variants.sh "--method 6 --reference-order --record none --steps 3" $A/librtmi_base.so $A/librtmi_inl.so
tools/ab_variants.sh "--method 9 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_inl.so
tools/ab_variants.sh "--scenario anisotropy --record none --steps 3" $A/librtmi_base.so $A/librtmi_inl.so
tools/ab_variants.sh "--scenario interface --method 9 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_inl.so
tools/ab_variants.sh "--scenario interface --method 4 --record none --steps 3" $A/librtmi_base.so $A/librtmi_inl.so
} > gpurun_out/r4_c10/ab.txt 2>&1
cat gpurun_out/r4_c10/ab.txt
