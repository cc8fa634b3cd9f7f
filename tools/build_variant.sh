#!/bin/bash
# Diagnostic: build build/variants/lib_<name>.so = the shipped library with ONE source recompiled with extra flags
#   tools/build_variant.sh <name> <source.hip> "<extra hipcc flags>"        (runs without a GPU)
# A/B the variants on one box with PARADIS_HIP_LIB=build/variants/lib_<name>.so (tools/adv_trace*.sh, tools/ab_libs.sh).
set -e
NAME=$1; SRC=$2; FLAGS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/build/variants; mkdir -p $OUT
BASE=$(basename $SRC .hip)
EXTRA="-fno-slp-vectorize"
[ $BASE = advect ] && EXTRA="$EXTRA -ffp-contract=off"
[ $BASE = feed ] && EXTRA="$EXTRA -ffp-contract=off"
OBJS=$(ls $R/build/obj/*.o | grep -v "/$BASE.o")
[ $BASE = gemm ] && OBJS=$(echo "$OBJS" | grep -v "/gemm_amp_")     # (gemm.hip without -DGEMM_PART holds both halves)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics $EXTRA $FLAGS \
    -c $R/paradis_model_amd/csrc/$BASE.hip -o $OUT/${BASE}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$NAME.so $OBJS $OUT/${BASE}_$NAME.o
echo "built $OUT/lib_$NAME.so"
