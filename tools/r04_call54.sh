#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4_c54
tools/profile_config.sh r04q_vert_full_plain --record full --mode plain > gpurun_out/r4_c54/profile.log 2>&1
tools/profile_config.sh r04q_vert_full --record full >> gpurun_out/r4_c54/profile.log 2>&1
python3 bench.py --mode plain > gpurun_out/r4_c54/bench_plain.json 2> gpurun_out/r4_c54/bench_plain.err; cut -c1-150 gpurun_out/r4_c54/bench_plain.json
echo done
