V=build/variants
for cfg in "--steps 10" "--record none --steps 10" "--scenario fisheye --record none --steps 10" "--rays 65536 --record none --steps 20" "--dtype f32 --rays 8388608 --record none --steps 5"; do
  echo "### $cfg"
  bash tools/ab_variants.sh "$cfg" $V/librtmi_head.so raytracing_amd/librtmi.so $V/librtmi_box.so
done
