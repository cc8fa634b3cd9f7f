#!/bin/bash
# Diagnostic: ChannelNorm backward with its two passes run over chunks of the batch (PARADIS_NORM_BWD_CHUNK_MB = read set
# of one chunk; 0 = one chunk) - in isolation (tools/norm_bwd_bench.py) and in the 32x64 B = 32 training step, one box.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/r6
OUT=gpurun_out/r6/norm_chunk_sweep.txt; : > $OUT
for mb in 0 24 48 96 160 0; do
  echo "== PARADIS_NORM_BWD_CHUNK_MB=$mb" >> $OUT
  PARADIS_NORM_BWD_CHUNK_MB=$mb python3 tools/norm_bwd_bench.py 2>&1 | grep "us " >> $OUT
done
for mb in 0 48 96 0 48 96; do
  echo "== step, PARADIS_NORM_BWD_CHUNK_MB=$mb" >> $OUT
  PARADIS_NORM_BWD_CHUNK_MB=$mb python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra-legs --no-other-configs 2>/dev/null \
    | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print(r['ms_per_step'], 'ms per step')" >> $OUT
done
cat $OUT
