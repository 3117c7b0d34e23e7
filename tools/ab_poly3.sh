V=build/variants
for cfg in "--scenario interface --record none --steps 5 --mode plain" "--record none --steps 10 --mode plain" "--steps 10 --mode plain" "--scenario fisheye --record none --steps 10 --mode plain" "--dtype f32 --rays 8388608 --record none --steps 5 --mode plain"; do
  echo "### $cfg"
  bash tools/ab_variants.sh "$cfg" raytracing_amd/librtmi.so $V/librtmi_g2.so $V/librtmi_g4.so $V/librtmi_g8.so
done
