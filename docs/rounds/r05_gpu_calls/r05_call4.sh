#!/bin/bash
O=gpurun_out/r5_c4; mkdir -p $O
echo "== default build" | tee $O/probe.txt
timeout -k 10 300 python tools/ref_small_probe.py 64 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
echo "== RTMI_NO_FLAT=1 (the flat path never taken)" | tee -a $O/probe.txt
RTMI_NO_FLAT=1 timeout -k 10 300 python tools/ref_small_probe.py 64 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
