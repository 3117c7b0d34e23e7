#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c21
A=build/ab
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4_c21/pytest.log 2>&1; echo "pytest (final build) rc=$?"; tail -2 gpurun_out/r4_c21/pytest.log
{
echo "## op7inl: op7's default (the reference-order step on the fast field lookup) with its sin/cos inline on the LDS table"
tools/ab_variants.sh "--method 7 --record none --steps 5" $A/librtmi_base.so $A/librtmi_op7inl.so
tools/ab_variants.sh "--scenario interface --method 7 --record none --steps 5" $A/librtmi_base.so $A/librtmi_op7inl.so
tools/ab_variants.sh "--scenario fisheye --method 7 --record none --steps 5" $A/librtmi_base.so $A/librtmi_op7inl.so
tools/ab_variants.sh "--method 7 --steps 5" $A/librtmi_base.so $A/librtmi_op7inl.so
} > gpurun_out/r4_c21/ab.txt 2>&1
cat gpurun_out/r4_c21/ab.txt | awk -F'  +' '{print $1" | "$2" | "$3}' | cut -c1-170
