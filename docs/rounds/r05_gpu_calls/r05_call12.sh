#!/bin/bash
# round 5, GPU call 12: k_advance_lat's kept-cell reload through the scalar cache (b) and plain row stores (c) against the default build
O=gpurun_out/r5_c12; mkdir -p $O
L="raytracing_amd/librtmi.so raytracing_amd/librtmi_b.so raytracing_amd/librtmi_c.so"
{
for rep in 1 2; do
tools/ab_libs.sh "$L" --total-rays 1048576 --emulate-world 8 --steps 20
tools/ab_libs.sh "$L" --total-rays 1048576 --emulate-world 8 --record none --steps 20
tools/ab_libs.sh "$L" --rays 65536 --steps 20
tools/ab_libs.sh "$L" --rays 65536 --record none --steps 20
done
tools/ab_libs.sh "$L" --total-rays 1048576 --emulate-world 4 --steps 20
tools/ab_libs.sh "$L" --scenario interface --total-rays 1048576 --emulate-world 8 --record none --steps 10
} 2>&1 | tee $O/ab_lat_reload.txt
