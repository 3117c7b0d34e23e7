V=build/variants
for cfg in "--steps 10" "--record none --steps 10" "--scenario fisheye --record none --steps 10" "--scenario interface --record none --steps 5" "--rays 65536 --record none --steps 20" "--dtype f32 --rays 8388608 --record none --steps 5" "--method 1 --record none --steps 5"; do
  echo "### $cfg"
  bash tools/ab_variants.sh "$cfg" $V/librtmi_base.so $V/librtmi_tiny.so $V/librtmi_chord.so raytracing_amd/librtmi.so
done
