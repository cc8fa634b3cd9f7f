#!/bin/bash
# Diagnostic: PMC counters for the pointwise GEMM (run on the GPU box from the repo root).
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|TCC_[A-Z_0-9]+(_sum)?|FETCH_SIZE|WRITE_SIZE|MfmaUtil|LDSBankConflict)\b" | sort -u > $R/gpurun_out/counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVES"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/gemm_bench.py 16,4 > $R/gpurun_out/pmc_$tag.log 2>&1
done
ls $R/gpurun_out/pmc_*/*/ | head
