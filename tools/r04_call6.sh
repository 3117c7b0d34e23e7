#!/bin/bash
# round 4, GPU call 6: the two full-size row-level tests (cfg3, cfg5); A/B of two build variants in one session
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c6
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "cfg5_anisotropy_full or cfg3_fisheye_full or headline" -s > gpurun_out/r4_c6/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r4_c6/pytest.log
A=build/ab
{
echo "## ftag: the wave keeps its last flat-map entry in scalar registers"
tools/ab_variants.sh "--scenario interface --record none --steps 5" $A/librtmi_base.so $A/librtmi_ftag.so
tools/ab_variants.sh "--record none --steps 10" $A/librtmi_base.so $A/librtmi_ftag.so
tools/ab_variants.sh "--steps 10" $A/librtmi_base.so $A/librtmi_ftag.so
tools/ab_variants.sh "--scenario fisheye --record none --steps 10" $A/librtmi_base.so $A/librtmi_ftag.so
echo "## gw2: the golden-section kernels built for two waves per SIMD (no spills) instead of three"
tools/ab_variants.sh "--scenario anisotropy --record none --steps 3" $A/librtmi_base.so $A/librtmi_gw2.so
tools/ab_variants.sh "--scenario interface --method 9 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_gw2.so
tools/ab_variants.sh "--scenario vert_heterogeneous --method 9 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_gw2.so
tools/ab_variants.sh "--scenario anisotropy --method 10 --rays 524288 --record none --steps 3" $A/librtmi_base.so $A/librtmi_gw2.so
} > gpurun_out/r4_c6/ab.txt 2>&1
cat gpurun_out/r4_c6/ab.txt
