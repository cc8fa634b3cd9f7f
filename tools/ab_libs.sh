#!/bin/bash
# Diagnostic: A/B of library builds on ONE box at the level of the training step (box-to-box spread ~4 %).
#   tools/ab_libs.sh "<variant names, 'shipped' = the in-tree library>" [rounds] [extra bench args]
# Variants are build/variants/lib_<name>.so (tools/build_variant.sh).  Rounds rotate the order of the variants.
NAMES=($1); ROUNDS=${2:-3}; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd)
n=${#NAMES[@]}
for i in $(seq 0 $((ROUNDS - 1))); do
  for j in $(seq 0 $((n - 1))); do
    v=${NAMES[$(((i + j) % n))]}
    LIB=""; [ "$v" != shipped ] && LIB=$R/build/variants/lib_$v.so
    PARADIS_HIP_LIB=$LIB python3 $R/bench.py --no-cpu-baseline --no-exact-leg --no-kernel-events --steps 10 --warmup 4 "$@" 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms')"
  done
done
