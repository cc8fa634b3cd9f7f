#!/bin/bash
# Diagnostic (GPU box): per-kernel times of the advection kernels inside the model, configs[3] and [4]
#   tools/adv_kernel_times.sh <tag>
TAG=$1
R=$PWD
for w in era5_1.4deg_128x256_S1_B8 era5_0.25deg_721x1440_fwd_B1; do
  tools/profile_cmd.sh ${TAG}_$w --workload $w --no-cpu-baseline --no-extra-legs --no-kernel-events --steps 3 --warmup 1
  f=$(ls -t $R/gpurun_out/prof_${TAG}_$w/stats/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -z "$f" ] && f=$(find $R/gpurun_out/prof_${TAG}_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("advect", "pole_row", "absmax", "fixed_to_float", "zero")):
        print("%-60s calls %5s avg %9.1f us total %8.2f ms" % (n.replace("(anonymous namespace)::", "").split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("all kernels: %.1f ms" % (tot / 1e6))
PY
done
