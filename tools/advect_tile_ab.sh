#!/bin/bash
# Diagnostic: tile shapes of the tiled advection schedule; args = compiler flag sets, one build each
# e.g. tools/advect_tile_ab.sh "-DADV_TILE_HB=16 -DADV_THREADS_B=512" "-DADV_TILE_HB=32 -DADV_THREADS_B=1024"
for v in "${@:-"-DADV_TILE_HF=64"}"; do
  echo "== $v"
  touch paradis_model_amd/csrc/advect.hip
  make -C paradis_model_amd/csrc FLAGS_advect="-ffp-contract=off $v" > /dev/null 2>&1 || echo BUILD FAILED
  for wl in era5_1.4deg_128x256_S1_B8; do
    python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['ms_per_step'], d['roofline_advect_fwd']['avg_launch_ms'], d['roofline_advect_bwd']['avg_launch_ms'])"
  done
done
touch paradis_model_amd/csrc/advect.hip; make -C paradis_model_amd/csrc > /dev/null 2>&1
