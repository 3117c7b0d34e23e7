#!/bin/bash
# round 4, GPU call 7: cfg5 variant (carried moments + predicted phase-T centre): bits, then A/B
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c7
A=build/ab
RTMI_LIB_PATH=$A/librtmi_pred.so timeout -k 10 900 python3 -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -m gpu -q -x -k "aniso or cfg5 or op10 or op11 or extreme or 10 or 11" > gpurun_out/r4_c7/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r4_c7/pytest.log
{
tools/ab_variants.sh "--scenario anisotropy --record none --steps 3" $A/librtmi_base.so $A/librtmi_pred.so
tools/ab_variants.sh "--scenario anisotropy --method 10 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_pred.so
tools/ab_variants.sh "--scenario anisotropy --record none --steps 5 --total-rays 1048576 --emulate-world 8" $A/librtmi_base.so $A/librtmi_pred.so
} > gpurun_out/r4_c7/ab.txt 2>&1
cat gpurun_out/r4_c7/ab.txt
