mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_hip_pad_advect.py tests/test_hip_amp.py -q -x 2>&1 | grep -E "^E   |passed|failed" | cut -c1-200 | head
python bench.py --steps 20 --warmup 5 > gpurun_out/r5/bench2.json 2> gpurun_out/r5/bench2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench2.json'))
print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'adv', d['roofline_advect_fwd']['frac'], d['roofline_advect_bwd']['frac'])
print('amp', d.get('bf16_mixed_amp'))
print('graph', d['hip_graph_replay']['ms_per_step'], d['hip_graph_replay']['host_ms_one_eager_step'])
for n,r in d['other_configs'].items():
    print(n, r.get('ms_per_step'), r.get('peak_hbm_gb'), {k:(r[k]['frac'], r[k]['avg_launch_ms']) for k in r if k.startswith('roofline')}, r.get('error'))
PY
python tools/advect_bench.py 0.05 2>&1 | grep -v "generic\|amdgpu\|32x64"
