mkdir -p gpurun_out/r3p_d
timeout -k 10 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -m gpu -q -x -k "exact or oracles_bits or cfg5 or aniso or sharding or cfg2 or checkpoint or within_1e9 or reference_order" > gpurun_out/r3p_d/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3p_d/pytest.log
python3 tools/lat_probe.py
RTMI_LAT_ONE_SLOT=1 python3 tools/lat_probe.py
for cfg in "--scenario anisotropy --record none --steps 3" "--method 9 --rays 524288 --record none --steps 3" "--scenario anisotropy --method 10 --rays 524288 --record none --steps 3" "--rays 65536 --steps 20" "--rays 65536 --record none --steps 20" "--total-rays 1048576 --emulate-world 8 --steps 10" "--total-rays 1048576 --emulate-world 8 --record none --steps 10" "--scenario anisotropy --record none --total-rays 1048576 --emulate-world 8 --steps 3"; do python3 tools/bench_line.py $cfg; done
