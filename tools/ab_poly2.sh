V=build/variants
for cfg in "--steps 10" "--record none --steps 10" "--scenario interface --record none --steps 5" "--scenario fisheye --record none --steps 10" "--scenario fisheye --steps 10" "--dtype f32 --rays 8388608 --record none --steps 5" "--method 1 --record none --steps 5"; do
  echo "### $cfg"
  bash tools/ab_variants.sh "$cfg" raytracing_amd/librtmi.so $V/librtmi_b1w5.so
done
echo "### latency build"
for cfg in "--rays 65536 --record none" "--rays 65536" "--total-rays 1048576 --emulate-world 8" "--total-rays 1048576 --emulate-world 8 --record none" "--total-rays 1048576 --emulate-world 4 --record none" "--total-rays 1048576 --emulate-world 4"; do
  for lat in 0 1; do
    if [ $lat = 0 ]; then export RTMI_NO_LAT=1; else unset RTMI_NO_LAT; fi
    echo -n "lat=$lat [$cfg] : "; python3 tools/bench_line.py $cfg --steps 10
  done
done
