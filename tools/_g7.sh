mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_hip_pad_advect.py tests/test_hip_determinism.py -q -x 2>&1 | grep -E "^E   |passed|failed" | cut -c1-200 | head
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5/bench3.json 2> gpurun_out/r5/bench3.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench3.json'))
print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'adv', d['roofline_advect_fwd']['frac'], d['roofline_advect_bwd']['frac'])
for n,r in d['other_configs'].items():
    print(n, r.get('ms_per_step'), {k:(round(r[k]['frac'],3), round(r[k]['avg_launch_ms'],3)) for k in r if k.startswith('roofline')}, r.get('error'))
PY
python tools/advect_bench.py 0.05 2>&1 | grep -v "generic\|amdgpu\|32x64"
