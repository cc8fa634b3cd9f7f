#!/bin/bash
# PARADIS_GEMM_STAGGER sweep (units of 512 cycles): isolated bf16-mixed GEMM launches, the bf16-mixed step and the fp32 step
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6/stagger
mkdir -p $O
for st in ${STAGGERS:-0 64 128 192 256 0}; do
  export PARADIS_GEMM_STAGGER=$st
  python3 $R/tools/gemm16_bench.py --shapes 896x896 --reps 30 2>/dev/null | grep -E "^896x896 +fwd" | awk -v s=$st '{print "stagger",s,$0}'
  for mode in "--amp" ""; do
    python3 $R/bench.py $mode --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --no-extra-legs --no-kernel-events > $O/step_${st}${mode}.json 2>/dev/null
    python3 -c "import json,sys; d=json.load(open('$O/step_${st}${mode}.json')); print('stagger $st step${mode:- fp32}: %.2f ms' % d['ms_per_step'])"
  done
done
