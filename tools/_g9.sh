mkdir -p gpurun_out/r5
tools/kernel_stats.sh 'dwconv' --workload era5_1.4deg_128x256_S1_B8
PARADIS_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/lib_notiles.so tools/kernel_stats.sh 'dwconv' --workload era5_1.4deg_128x256_S1_B8
timeout 900 python -m pytest tests/test_hip_gemm_split.py tests/test_hip_amp.py -q -x 2>&1 | grep -E "^E   |passed|failed|MEASURED" | cut -c1-250 | head -8
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r5/bench6.json 2> gpurun_out/r5/bench6.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench6.json'))
print('headline', d['value'], d['ms_per_step'])
print('amp', {k:v for k,v in d['bf16_mixed_amp'].items() if k!='gemm_arithmetic'})
PY
