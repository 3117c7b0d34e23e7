V=build/variants
for cfg in "--scenario anisotropy --record none --steps 3" "--scenario anisotropy --method 10 --rays 524288 --record none --steps 3"; do
  echo "### $cfg"
  bash tools/ab_variants.sh "$cfg" raytracing_amd/librtmi.so $V/librtmi_t11.so $V/librtmi_t10.so $V/librtmi_t9.so
done
