#!/bin/bash
# Usage: tools/profile_cmd.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/stats (kernel trace + stats)
TAG=$1; shift
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/stats -- python3 $R/bench.py "$@" > $R/gpurun_out/prof_$TAG.stats.log 2>&1
