V=build/variants
for cfg in "--scenario interface --record none --steps 5" "--scenario interface --steps 5 --rec-rows 4100" "--steps 10" "--record none --steps 10" "--scenario fisheye --record none --steps 10" "--dtype f32 --rays 8388608 --record none --steps 5" "--total-rays 1048576 --emulate-world 4 --record none --steps 10"; do
  echo "### $cfg"
  bash tools/ab_variants.sh "$cfg" $V/librtmi_nopf.so raytracing_amd/librtmi.so
done
