#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c11
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > gpurun_out/r4_c11/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r4_c11/pytest.log
A=build/ab
{
echo "## new: inline sin/cos for op3/4/5/9/10/11 only"
tools/ab_variants.sh "--method 7 --record none --steps 3" $A/librtmi_base.so $A/librtmi_new.so
tools/ab_variants.sh "--method 3 --record none --steps 3" $A/librtmi_base.so $A/librtmi_new.so
tools/ab_variants.sh "--method 6 --reference-order --record none --steps 3" $A/librtmi_base.so $A/librtmi_new.so
tools/ab_variants.sh "--method 9 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_new.so
tools/ab_variants.sh "--scenario anisotropy --record none --steps 3" $A/librtmi_base.so $A/librtmi_new.so
tools/ab_variants.sh "--scenario interface --method 9 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_new.so
tools/ab_variants.sh "--scenario interface --method 5 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_new.so
} > gpurun_out/r4_c11/ab.txt 2>&1
cat gpurun_out/r4_c11/ab.txt | awk -F'  +' '{print $1" | "$2" | "$3}' | cut -c1-170
