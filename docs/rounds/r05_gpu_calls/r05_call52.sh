#!/bin/bash
# round 5, GPU call 52: do the register-bound kernels like another kernel-argument layout better?  BatchDev padded by 8 / 16 / 24 bytes behind Consts (RTMI_X_PAD)
O=gpurun_out/r5_c52; mkdir -p $O
V=build/variants
{
for args in "--scenario anisotropy --record none --steps 3 --mode sliced" "--scenario interface --method 9 --rays 524288 --record none --steps 3 --mode plain" "--steps 10 --mode sliced" "--scenario interface --record none --steps 10 --mode plain" "--method 3 --record none --steps 3 --mode plain" "--method 9 --rays 524288 --record none --steps 3 --mode plain"; do
echo "-- $args"
tools/ab_variants.sh "$args" raytracing_amd/librtmi.so $V/librtmi_pad1.so $V/librtmi_pad2.so $V/librtmi_pad3.so
done
} 2>&1 | tee $O/ab.txt
