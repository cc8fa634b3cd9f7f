#!/bin/bash
# FETCH_SIZE per launch of the split weight-gradient kernel: tools/wgrad_traffic.sh <tag> [H W B]   (GPU box, repo root)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=$1; shift
mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/wg_$T -o out -- python3 $R/tools/wgrad_traffic.py "$@" > $R/gpurun_out/wg_$T.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/wg_$T/**/out_counter_collection.csv", recursive=True)
k = glob.glob("$R/gpurun_out/wg_$T/**/out_kernel_trace.csv", recursive=True)
if not f or not k: print("no counters"); raise SystemExit
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(k[0]))}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    ns, name = dur.get(r["Dispatch_Id"], (0, ""))
    if "wgrad_split" in name: acc[r["Grid_Size"]][r["Counter_Name"]].append((float(r["Counter_Value"]), ns))
for grid, c in acc.items():
    fs = sorted(v for v, _ in c["FETCH_SIZE"]); n = len(fs)
    clk = sorted(v / 8.0 / ns for v, ns in c["GRBM_GUI_ACTIVE"])[n // 2]
    us = sorted(ns for _, ns in c["FETCH_SIZE"])[n // 2] / 1e3
    # FETCH_SIZE is in KB on gfx950 and counts 64 B per 128-B request: x 2 (MI355X_MICROARCH.md)
    print("$T grid %s: %d launches, median %.1f us, FETCH_SIZE x2 = %.2f GB per launch, clock %.2f GHz" % (grid, n, us, 2 * fs[n // 2] * 1024 / 1e9, clk))
PY
