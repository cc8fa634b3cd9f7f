#!/bin/bash
# Usage: tools/advect_pmc.sh [script args...]   (default script tools/advect_pmc.py; set PMC_SCRIPT to override)
R=$PWD; cd /tmp; export TMPDIR=/tmp
SCRIPT=${PMC_SCRIPT:-tools/advect_pmc.py}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_VMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/apmc_$i -- python3 $R/$SCRIPT "$@" > $R/gpurun_out/apmc_$i.log 2>&1
done
