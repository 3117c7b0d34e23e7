#!/bin/bash
# round 5, GPU call 37: (a) the headline against the build before today's re-trace work (same box); (b) four scheduler-flag builds on the register-bound kernels
O=gpurun_out/r5_c37; mkdir -p $O
V=build/variants
{
echo "== headline, sliced and plain: old602 (commit 602c303) vs now"
tools/ab_variants.sh "--steps 10 --mode sliced" $V/librtmi_old602.so raytracing_amd/librtmi.so
tools/ab_variants.sh "--steps 10 --mode plain" $V/librtmi_old602.so raytracing_amd/librtmi.so
echo "== scheduler flags"
for args in "--scenario anisotropy --record none --steps 3 --mode sliced" "--scenario interface --method 9 --rays 524288 --record none --steps 3 --mode plain" "--steps 10 --mode sliced" "--scenario interface --record none --steps 10 --mode plain"; do
echo "-- $args"
tools/ab_variants.sh "$args" raytracing_amd/librtmi.so $V/librtmi_maxilp.so $V/librtmi_bias0.so $V/librtmi_trackers.so $V/librtmi_nohighrp.so
done
} 2>&1 | tee $O/ab.txt
