#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run from the repo root):
#   1. --kernel-trace --stats of the default bench command  -> gpurun_out/prof_$TAG/stats
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (TCC slots; MI355X_MICROARCH.md)
#   3. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (matrix-pipe utilisation per kernel)
# Summaries are then written by tools/summarize_profile.py into profiles/.
TAG=${1:-r02}
R=$PWD; cd /tmp; export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-exact-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/stats -- $CMD > $R/gpurun_out/prof_$TAG.stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/fetch -- $CMD > $R/gpurun_out/prof_$TAG.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$TAG/write -- $CMD > $R/gpurun_out/prof_$TAG.write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_$TAG/mfma -- $CMD > $R/gpurun_out/prof_$TAG.mfma.log 2>&1
cd $R; python3 tools/summarize_profile.py $TAG
