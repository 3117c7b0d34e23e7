#!/bin/bash
set -u
tools/profile_config.sh r03q_cfg5_aniso_none --scenario anisotropy --record none
tools/profile_config.sh r03q_vert_full --record full
tools/profile_config.sh r03q_vert_none --record none
tools/profile_config.sh r03q_cfg3_fisheye_none --scenario fisheye --record none
tools/profile_config.sh r03q_cfg4_f32_none --dtype f32 --rays 8388608 --record none
tools/profile_config.sh r03q_cfg2_none --rays 65536 --record none
