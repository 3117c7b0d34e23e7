#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c16
A=build/ab
{
echo "## fw5: the flat-capable k_advance (light methods) built for five waves per SIMD (96 VGPRs, 8 spilled) instead of four"
tools/ab_variants.sh "--scenario interface --record none --steps 5" $A/librtmi_base.so $A/librtmi_fw5.so
tools/ab_variants.sh "--scenario interface --steps 5 --rec-rows 4100" $A/librtmi_base.so $A/librtmi_fw5.so
tools/ab_variants.sh "--scenario interface --method 2 --record none --steps 5" $A/librtmi_base.so $A/librtmi_fw5.so
tools/ab_variants.sh "--scenario interface --method 8 --record none --steps 5" $A/librtmi_base.so $A/librtmi_fw5.so
tools/ab_variants.sh "--scenario interface --method 7 --record none --steps 5" $A/librtmi_base.so $A/librtmi_fw5.so
} > gpurun_out/r4_c16/ab.txt 2>&1
cat gpurun_out/r4_c16/ab.txt | awk -F'  +' '{print $1" | "$2" | "$3}' | cut -c1-170
