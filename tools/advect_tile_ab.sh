#!/bin/bash
# Diagnostic: forward tile height of the tiled advection schedule (rebuilds the library per variant)
for v in 16 32 64; do
  echo "== TILE_HF=$v"
  touch paradis_model_amd/csrc/advect.hip
  make -C paradis_model_amd/csrc FLAGS_advect="-ffp-contract=off -DADV_TILE_HF=$v" > /dev/null 2>&1 || echo BUILD FAILED
  for wl in era5_1.4deg_128x256_S1_B8 era5_0.25deg_721x1440_fwd_B1; do
    python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'], d['ms_per_step'], d['roofline_advect_fwd']['avg_launch_ms'])"
  done
done
touch paradis_model_amd/csrc/advect.hip; make -C paradis_model_amd/csrc > /dev/null 2>&1
