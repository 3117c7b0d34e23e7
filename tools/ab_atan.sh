V=build/variants
python -m pytest tests/test_gpu_parity.py tests/test_gpu_exact.py -m gpu -q -x 2>&1 | tail -3
for cfg in "--method 1 --record none --steps 5" "--method 8 --record none --steps 5" "--method 7 --record none --steps 5" "--method 1 --scenario fisheye --record none --steps 5" "--method 8 --scenario interface --record none --steps 3" "--method 1 --steps 5"; do
  echo "### $cfg"
  bash tools/ab_variants.sh "$cfg" $V/librtmi_at0.so raytracing_amd/librtmi.so
done
