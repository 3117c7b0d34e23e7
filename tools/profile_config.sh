#!/bin/bash
# Profile one bench.py configuration with rocprofv3 on the GPU box (run through gpurun):
#   tools/profile_config.sh <tag> <bench.py args...>
# Passes (separate runs, as MI355X_MICROARCH.md prescribes: --pmc never together with tracing; FETCH_SIZE and WRITE_SIZE
# do not fit one pass): kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ instruction/cycle counters, fp64 mix + clock.
# Raw output under gpurun_out/prof_<tag>/; tools/profile_summary.py turns it into profiles/<tag>_*.
set -u
tag="$1"; shift
out="gpurun_out/prof_${tag}"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--steps 3 --warmup 1 --cpu-seconds 0 $*"
python3 bench.py $args > "$out/bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- python3 bench.py $args > "$out/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o run -- python3 bench.py $args > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o run -- python3 bench.py $args > "$out/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
  --output-format csv -d "$out/pmc_sq" -o run -- python3 bench.py $args > "$out/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d "$out/pmc_mix" -o run -- python3 bench.py $args > "$out/pmc_mix.log" 2>&1
echo "$args" > "$out/args.txt"
ls "$out"
