#!/bin/bash
# usage: tools/adv_trace_bwd.sh <variant names...>  -> median duration of the W=64 backward advection kernel per
# build/variants/lib_<name>.so (rocprofv3 --kernel-trace; GPU box, repo root)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  rm -rf /tmp/trb_$n; PARADIS_HIP_LIB=$R/build/variants/lib_$n.so rocprofv3 --kernel-trace -d /tmp/trb_$n -o out --output-format csv -- python3 $R/tools/adv_bwd_only.py 0.05 > /dev/null 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("/tmp/trb_$n/out_kernel_trace.csv")) if "bwd_row64" in r["Kernel_Name"]]
d=sorted(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows[-60:])
print("%-10s median %.1f us  min %.1f us  (n=%d)" % ("$n", d[len(d)//2]/1e3, d[0]/1e3, len(d)))
PY
done
