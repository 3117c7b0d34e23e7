#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c3
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > gpurun_out/r4_c3/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r4_c3/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
