#!/bin/bash
# MFMA busy / clock / issue counters of the split forward GEMM: tools/gemm_pmc.sh <f16x2|bf16x3> <tag>  (GPU box, repo root)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r2/gpmc_$2_$i -o out -- python3 $R/tools/split_gemm_pmc.py $1 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/r2/gpmc_$2_*")):
    try:
        rows = list(csv.DictReader(open(d + "/out_counter_collection.csv")))
    except OSError:
        print(d, "no counters"); continue
    acc = collections.defaultdict(list)
    for r in rows:
        if "pw_gemm_split" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    kt = [r for r in csv.DictReader(open(d + "/out_kernel_trace.csv")) if "pw_gemm_split" in r["Kernel_Name"]]
    dur = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt[-50:])
    print(d.split("/")[-1], "median us", dur[len(dur)//2] / 1e3)
    for k, v in acc.items():
        v = sorted(v[-50:]); print("   ", k, v[len(v)//2])
PY
