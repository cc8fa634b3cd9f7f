#!/bin/bash
# Diagnostic (GPU box): parity tests of the advection kernels, isolated bench at the two large grids, in-model bench
# lines for configs[3] / [4]:   tools/adv_strip_check.sh <tag>
TAG=$1
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_hip_pad_advect.py tests/test_hip_determinism.py -x -q -m gpu 2>&1 | tail -5
ADVECT_BENCH_ONLY=128x256 timeout 300 python tools/advect_bench.py 0.05 0.5 2>&1 | grep bicubic
ADVECT_BENCH_ONLY=721x1440 timeout 300 python tools/advect_bench.py 0.05 0.5 2>&1 | grep bicubic
for w in era5_1.4deg_128x256_S1_B8 era5_0.25deg_721x1440_fwd_B1; do
  python bench.py --workload $w --no-cpu-baseline --no-extra-legs --steps 5 --warmup 2 2>gpurun_out/r4/err_${TAG}_$w.txt | tail -1 > gpurun_out/r4/bench_${TAG}_$w.json
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r4/bench_${TAG}_$w.json"))
    print("$w", round(d["ms_per_step"], 1), "ms/step", {k: (round(d[k]["frac"], 3), round(d[k]["avg_launch_ms"], 3)) for k in d if k.startswith("roofline")})
except Exception as e:
    print("$w failed", e); print(open("gpurun_out/r4/err_${TAG}_$w.txt").read()[-1500:])
PY
done
