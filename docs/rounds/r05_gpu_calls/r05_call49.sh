#!/bin/bash
# round 5, GPU call 49: the next cell's loads issued BEFORE the step's row stores (in-order memory counter): few-waves configurations, A/B against the build before
O=gpurun_out/r5_c49; mkdir -p $O
V=build/variants
{
for args in "--emulate-world 8 --record full --steps 10" "--emulate-world 8 --record none --steps 10" "--rays 65536 --record full --steps 10" "--rays 65536 --record none --steps 10" "--scenario fisheye --emulate-world 8 --record full --steps 10" "--scenario interface --emulate-world 8 --record full --rec-rows 4100 --steps 10"; do
echo "-- $args"
tools/ab_variants.sh "$args" $V/librtmi_late.so raytracing_amd/librtmi.so
done
} 2>&1 | tee $O/ab.txt
