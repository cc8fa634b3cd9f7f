#!/bin/bash
# usage: tools/adv_trace.sh <variant names...>  -> median kernel duration of the W=64 forward advection kernel per
# build/variants/lib_<name>.so (rocprofv3 --kernel-trace; run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  for sc in 0.05; do
    rm -rf /tmp/tr_$n; PARADIS_HIP_LIB=$R/build/variants/lib_$n.so rocprofv3 --kernel-trace -d /tmp/tr_$n -o out --output-format csv -- python3 $R/tools/adv_fwd_only.py $sc > /dev/null 2>&1
    python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("/tmp/tr_$n/out_kernel_trace.csv")) if "row64" in r["Kernel_Name"]]
d=sorted(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows[-100:])
print("%-16s scale $sc  median %.1f us  min %.1f us  (n=%d)" % ("$n", d[len(d)//2]/1e3, d[0]/1e3, len(d)))
PY
  done
done
