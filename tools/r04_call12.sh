#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c12
timeout -k 10 600 python3 -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_shard.py -m gpu -q -x -k "op7 or 7 or shard or reference_order" -s > gpurun_out/r4_c12/pytest.log 2>&1; echo "pytest rc=$?"
grep "op7 vs oracle" gpurun_out/r4_c12/pytest.log; tail -3 gpurun_out/r4_c12/pytest.log
timeout -k 10 600 python3 tools/parity_sweep.py --methods 7 > gpurun_out/r4_c12/sweep_op7.txt 2>&1
cat gpurun_out/r4_c12/sweep_op7.txt
{
python3 tools/bench_line.py --method 7 --record none --steps 3
python3 tools/bench_line.py --method 7 --record none --steps 3 --reference-order
python3 tools/bench_line.py --method 7 --record none --steps 3 --fused
python3 tools/bench_line.py --scenario interface --method 7 --record none --steps 3
python3 tools/bench_line.py --scenario interface --method 7 --record none --steps 3 --reference-order
python3 tools/bench_line.py --scenario fisheye --method 7 --record none --steps 3
python3 tools/bench_line.py --method 7 --steps 3
} > gpurun_out/r4_c12/rates.txt 2>&1
cat gpurun_out/r4_c12/rates.txt | cut -c1-200
