#!/bin/bash
# Diagnostic: A/B two builds of libparadis_hip.so on ONE box at the level of the training step
# (box-to-box spread is ~4 %, isolated-kernel A/Bs do not always carry over to the step).
#   tools/ab_step.sh "<extra hipcc flags for gemm.hip of build B>" [rounds]
# Run the build part locally (no GPU needed), the measuring part through gpurun.
FLAGS=$1; ROUNDS=${2:-3}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/build/ab; mkdir -p $OUT
if ! python3 -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)" 2>/dev/null; then
  OBJS=$(ls $R/build/obj/*.o | grep -v gemm.o)
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize $FLAGS \
      -c $R/paradis_model_amd/csrc/gemm.hip -o $OUT/gemm_b.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libparadis_hip_b.so $OBJS $OUT/gemm_b.o && echo built
  exit 0
fi
for i in $(seq $ROUNDS); do
  for v in A B; do
    LIB=""; [ $v = B ] && LIB=$OUT/libparadis_hip_b.so
    PARADIS_HIP_LIB=$LIB python3 $R/bench.py --no-cpu-baseline --no-exact-leg --no-kernel-events --steps 10 --warmup 3 2>/dev/null | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), 'samples/s', round(d['ms_per_step'],1), 'ms')"
  done
done
