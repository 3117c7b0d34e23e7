#!/bin/bash
# round 4, GPU call 1: HIP runtime probe (three import orders), baseline test suite, interface x op9 / op5 / op6 profiles
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c1
for m in rtmi_first torch_first preload; do
  timeout -k 10 300 python3 tools/hip_runtime_probe.py $m > gpurun_out/r4_c1/probe_$m.txt 2>&1; echo "probe $m rc=$?" >> gpurun_out/r4_c1/probe_$m.txt
done
echo "probes done"
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_c1/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r4_c1/pytest.log
tools/profile_config.sh r04_iface_op9_base --scenario interface --method 9 --rays 524288 --record none
tools/profile_config.sh r04_iface_op5_base --scenario interface --method 5 --rays 524288 --record none
echo done
