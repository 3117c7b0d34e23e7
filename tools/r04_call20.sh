#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c20
A=build/ab
RTMI_LIB_PATH=$PWD/$A/librtmi_ldstab.so timeout -k 10 600 python3 -m pytest tests/test_gpu_exact.py -m gpu -q -x > gpurun_out/r4_c20/pytest.log 2>&1; echo "pytest (exactness, LDS-table build) rc=$?"; tail -2 gpurun_out/r4_c20/pytest.log
{
echo "## ldstab: glibc's sin/cos table staged in LDS for the inline evaluations"
tools/ab_variants.sh "--scenario interface --method 9 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_ldstab.so
tools/ab_variants.sh "--method 9 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_ldstab.so
tools/ab_variants.sh "--method 3 --record none --steps 3" $A/librtmi_base.so $A/librtmi_ldstab.so
tools/ab_variants.sh "--scenario interface --method 4 --record none --steps 3" $A/librtmi_base.so $A/librtmi_ldstab.so
tools/ab_variants.sh "--scenario interface --method 5 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_ldstab.so
tools/ab_variants.sh "--scenario anisotropy --record none --steps 3" $A/librtmi_base.so $A/librtmi_ldstab.so
tools/ab_variants.sh "--scenario fisheye --method 5 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_ldstab.so
} > gpurun_out/r4_c20/ab.txt 2>&1
cat gpurun_out/r4_c20/ab.txt | awk -F'  +' '{print $1" | "$2" | "$3}' | cut -c1-170
