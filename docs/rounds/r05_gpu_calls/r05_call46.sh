#!/bin/bash
# round 5, GPU call 46: the few-waves kernel (k_advance_lat) with every lane's own kept cell and the next cell's loads a step ahead (rt::PolyLaneKept) against the wave's kept cell (lat0)
O=gpurun_out/r5_c46; mkdir -p $O
V=build/variants
{
for args in "--emulate-world 8 --record full --steps 10" "--emulate-world 8 --record none --steps 10" "--rays 65536 --record full --steps 10" "--rays 65536 --record none --steps 10" "--scenario interface --emulate-world 8 --record none --steps 10" "--scenario fisheye --emulate-world 8 --record none --steps 10" "--rays 32768 --record full --steps 10"; do
echo "-- $args"
tools/ab_variants.sh "$args" $V/librtmi_lat0.so raytracing_amd/librtmi.so
done
} 2>&1 | tee $O/ab.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/pytest.log
