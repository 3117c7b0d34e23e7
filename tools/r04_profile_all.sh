#!/bin/bash
# round-4 profile set of the present kernels (run through gpurun, in two halves: `tools/r04_profile_all.sh 1` and `... 2`);
# tools/profile_summary.py <tag>... turns gpurun_out/prof_<tag>/ into profiles/<tag>_* and profiles/traffic.json
set -u
half="${1:-1}"
if [ "$half" = 1 ]; then
tools/profile_config.sh r04q_vert_full --record full
tools/profile_config.sh r04q_vert_none --record none
tools/profile_config.sh r04q_cfg2_full --rays 65536 --record full
tools/profile_config.sh r04q_cfg2_none --rays 65536 --record none
tools/profile_config.sh r04q_cfg3_fisheye_none --scenario fisheye --record none
tools/profile_config.sh r04q_cfg3_fisheye_full --scenario fisheye --record full
tools/profile_config.sh r04q_cfg4_f32_none --dtype f32 --rays 8388608 --record none
tools/profile_config.sh r04q_iface_none --scenario interface --record none
else
tools/profile_config.sh r04q_cfg5_aniso_none --scenario anisotropy --record none
tools/profile_config.sh r04q_cfg5_shard8_none --scenario anisotropy --record none --total-rays 1048576 --emulate-world 8
tools/profile_config.sh r04q_iface_op9_none --scenario interface --method 9 --rays 524288 --record none
tools/profile_config.sh r04q_op7_vert_none --method 7 --record none
tools/profile_config.sh r04q_op7_fastfield_vert_none --method 7 --record none --fast-field
tools/profile_config.sh r04q_op3_vert_none --method 3 --record none
tools/profile_config.sh r04q_op9_vert_none --method 9 --rays 524288 --record none
tools/profile_config.sh r04q_strong8_full --total-rays 1048576 --emulate-world 8 --record full
fi
