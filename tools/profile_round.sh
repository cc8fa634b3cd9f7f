#!/bin/bash
# Collect rocprofv3 evidence on the GPU box (run from the repo root):
#   tools/profile_round.sh <TAG> [bench.py args ...]
#   1. --kernel-trace --stats of the bench command  -> gpurun_out/prof_$TAG/stats
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (TCC slots; MI355X_MICROARCH.md)
#   3. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (matrix-pipe utilisation per kernel)
# Summaries are then written by tools/summarize_profile.py into profiles/ (gpurun_out/profiles_out on the box).
# The program after `--` is python3 itself (no env / bash hop: the profiler's library initialises the GPU first).
TAG=${1:-r03}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
# the GPU box has no .git: `git rev-parse --short HEAD > .build_commit` before the gpurun call records the commit
export PARADIS_COMMIT=${PARADIS_COMMIT:-$(cat $R/.build_commit 2>/dev/null)}
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-legs $*"
mkdir -p $R/gpurun_out/prof_$TAG
echo "python3 bench.py $ARGS" > $R/gpurun_out/prof_$TAG/command.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/write -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_$TAG/mfma -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.mfma.log 2>&1
cd $R; PARADIS_PROFILE_DST=$R/gpurun_out/profiles_out python3 tools/summarize_profile.py $TAG
