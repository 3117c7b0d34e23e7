#!/bin/bash
# round-5 profile set (run through gpurun); tools/profile_summary.py <tag>... turns gpurun_out/prof_<tag>/ into profiles/<tag>_* and
# profiles/traffic.json
set -u
tools/profile_config.sh r05_vert_full --record full
tools/profile_config.sh r05_strong8_full --emulate-world 8 --record full
tools/profile_config.sh r05_strong8_none --emulate-world 8 --record none
tools/profile_config.sh r05_cfg2_full --rays 65536 --record full
tools/profile_config.sh r05_iface_none --scenario interface --record none
tools/profile_config.sh r05_iface_full --scenario interface --record full --rec-rows 4100
