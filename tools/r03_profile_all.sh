#!/bin/bash
# round-3 profile set (run through gpurun); tools/profile_summary.py <tag>... turns gpurun_out/prof_<tag>/ into profiles/<tag>_*
set -u
tools/profile_config.sh r03_vert_full_sliced --record full --mode sliced
tools/profile_config.sh r03_cfg2_full --rays 65536 --record full --mode plain
tools/profile_config.sh r03_cfg2_none --rays 65536 --record none --mode plain
tools/profile_config.sh r03_cfg5_aniso_none_sliced --scenario anisotropy --record none --mode sliced
tools/profile_config.sh r03_op9_vert_none_sliced --method 9 --rays 524288 --record none --mode sliced
tools/profile_config.sh r03_cfg3_fisheye_none_sliced --scenario fisheye --record none --mode sliced
tools/profile_config.sh r03_strong8_full --total-rays 1048576 --emulate-world 8 --record full --mode plain
