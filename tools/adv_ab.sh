#!/bin/bash
# usage (GPU box, repo root): tools/adv_ab.sh <variant names...>   ("shipped" = the in-tree library); two rounds, rotated
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in 1 2; do
  for n in "$@"; do
    LIB=$R/build/variants/lib_$n.so; [ $n = shipped ] && LIB=""
    PARADIS_HIP_LIB=$LIB python3 $R/tools/adv_ab.py 2>&1 | tail -1
  done
done
