#!/bin/bash
# round 4, GPU call 27: op7's default is reference order throughout again (fast-field step = RTMI_ORDER_FAST_FIELD): full GPU suite,
# the critical-ray window, op7's profiles, every method's rate, the sweep, the 1 M-ray interface fan of op7
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c27
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r4_c27/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_c27/pytest.log
timeout -k 10 600 python3 tools/critical_ray_window.py > gpurun_out/r4_c27/critical_ray_window.txt 2>&1; cat gpurun_out/r4_c27/critical_ray_window.txt
tools/profile_config.sh r04q_op7_vert_none --method 7 --record none > gpurun_out/r4_c27/profile.log 2>&1
tools/profile_config.sh r04q_op7_fastfield_vert_none --method 7 --record none --fast-field >> gpurun_out/r4_c27/profile.log 2>&1
echo "profiles done"
bash tools/all_methods_rate.sh > gpurun_out/r4_c27/all_methods_rate.txt 2>&1
echo "rates done"; grep "op7\|method 7" gpurun_out/r4_c27/all_methods_rate.txt
timeout -k 10 600 python3 tools/parity_sweep.py > gpurun_out/r4_c27/parity_sweep.txt 2>&1
tail -2 gpurun_out/r4_c27/parity_sweep.txt | cut -c1-400
python3 bench.py --scenario interface --method 7 --rays 1048576 --record stride:16 --rec-rows 600 --steps 2 --cpu-seconds 0 --parity-stride 64 2>/dev/null > gpurun_out/r4_c27/iface_1m_op7.json; cut -c1-600 gpurun_out/r4_c27/iface_1m_op7.json
