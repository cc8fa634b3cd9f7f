#!/bin/bash
# Per-kernel time of one short bench run (rocprofv3 --kernel-trace --stats), filtered by a name pattern:
#   tools/kernel_stats.sh 'dwconv|split_weights' [bench.py arguments]
PAT=${1:-.}
shift
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ks
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ks -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-legs "$@" > $R/gpurun_out/prof_ks.log 2>&1
cd $R
python3 - "$PAT" <<'PY'
import csv, glob, re, sys
f = glob.glob("gpurun_out/prof_ks/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
gemm = sum(float(r["TotalDurationNs"]) for r in rows if "pw_gemm" in r["Name"])
print("total kernel ms %.1f   GEMM %.1f   non-GEMM %.1f   (4 steps traced)" % (tot / 1e6, gemm / 1e6, (tot - gemm) / 1e6))
for r in rows:
    if re.search(sys.argv[1], r["Name"]):
        print("%-64s %5s calls %9.2f ms  avg %8.1f us" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
