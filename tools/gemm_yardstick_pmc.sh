#!/bin/bash
# Matrix-pipe busy share and effective clock of the vendor bf16 GEMM kernels next to the split kernels, same box:
#   tools/gemm_yardstick_pmc.sh <tag>      (GPU box, repo root; writes gpurun_out/yard_<tag>_pmc.txt)
# One counter pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_BUSY_CU_CYCLES) over `gemm_yardstick.py --short`.
# busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles);  clock = GRBM_GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out; D=$R/gpurun_out/yard_$1_pmc
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $D -o out \
  -- python3 $R/tools/gemm_yardstick.py --short --rounds 3 --reps 4 > $D.log 2>&1
python3 - <<PY | tee $R/gpurun_out/yard_$1_pmc.txt
import csv, glob, collections
d = "$D"
f = glob.glob(d + "/**/out_counter_collection.csv", recursive=True)
k = glob.glob(d + "/**/out_kernel_trace.csv", recursive=True)
if not f or not k:
    print("no counter output"); raise SystemExit
dur = {}
for r in csv.DictReader(open(k[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    did = r["Dispatch_Id"]
    if did not in dur: continue
    ns, name = dur[did]
    if ns < 100000: continue                      # GEMM launches only
    key = "ours:" + name.split("<")[0][:40] if "pw_gemm" in name else "vendor:" + name[:60]
    acc[key][r["Counter_Name"]].append((float(r["Counter_Value"]), ns))
print(f"{'kernel':70s} {'launches':>8s} {'median us':>10s} {'mfma busy':>10s} {'clock GHz':>10s}")
for key, c in sorted(acc.items()):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c: continue
    n = len(c["GRBM_GUI_ACTIVE"])
    clk = sorted(v / 8.0 / ns for v, ns in c["GRBM_GUI_ACTIVE"])[n // 2]                 # cycles per ns = GHz
    busy = sorted(v / (1024.0 * (g / 8.0)) for (v, ns), (g, _) in zip(c["SQ_VALU_MFMA_BUSY_CYCLES"], c["GRBM_GUI_ACTIVE"]))[n // 2]
    us = sorted(ns for _, ns in c["GRBM_GUI_ACTIVE"])[n // 2] / 1e3
    print(f"{key:70s} {n:8d} {us:10.1f} {busy:10.3f} {clk:10.2f}")
PY
