#!/bin/bash
# Diagnostic: rebuild the library with advect ablation macros and print tools/advect_accuracy.py for each
for v in "-DADV_IEEE_DIV" "-DADV_OCML_ATAN2" "-DADV_OCML_SINCOS" "-DADV_IEEE_DIV -DADV_OCML_ATAN2" ""; do
  echo "== variant: [$v]"
  touch paradis_model_amd/csrc/advect.hip
  make -C paradis_model_amd/csrc FLAGS_advect="-ffp-contract=off $v" > /dev/null 2>&1 || echo BUILD FAILED
  python tools/advect_accuracy.py 2>&1 | grep -E "^[0-9]+x"
done
