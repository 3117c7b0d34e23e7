V=build/variants
for cfg in "--rays 65536 --steps 20" "--rays 65536 --record none --steps 20" "--total-rays 1048576 --emulate-world 8 --steps 10" "--total-rays 1048576 --emulate-world 8 --record none --steps 10" "--rays 32768 --steps 20"; do
  echo "### $cfg"
  bash tools/ab_variants.sh "$cfg" $V/librtmi_kv.so raytracing_amd/librtmi.so
done
