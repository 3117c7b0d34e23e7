#!/bin/bash
# round 4, GPU call 2: full GPU suite with this round's new tests; timing of the inline tie path (interface x op9) and of op7's new default
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c2
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_c2/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r4_c2/pytest.log
{
python3 tools/bench_line.py --scenario interface --method 9 --rays 524288 --record none --steps 3
python3 tools/bench_line.py --scenario interface --method 5 --rays 524288 --record none --steps 3
python3 tools/bench_line.py --scenario vert_heterogeneous --method 9 --rays 524288 --record none --steps 3
python3 tools/bench_line.py --scenario fisheye --method 9 --rays 524288 --record none --steps 3
python3 tools/bench_line.py --scenario vert_heterogeneous --method 7 --record none --steps 3
python3 tools/bench_line.py --scenario vert_heterogeneous --method 7 --record none --steps 3 --fused
python3 tools/bench_line.py --scenario interface --method 7 --record none --steps 3
python3 tools/bench_line.py --scenario interface --method 7 --record none --steps 3 --fused
python3 tools/bench_line.py --scenario interface --method 6 --record none --steps 5
python3 tools/bench_line.py --scenario anisotropy --record none --steps 3
python3 tools/bench_line.py --steps 10
} > gpurun_out/r4_c2/rates.txt 2>&1
cat gpurun_out/r4_c2/rates.txt
