mkdir -p gpurun_out/r5
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_pad_advect.py -q -x 2>&1 | grep -E "^E   |passed|failed" | cut -c1-200 | head -5
cd /tmp; export TMPDIR=/tmp
for lib in r5 r4 r5 r4; do
  if [ $lib = r4 ]; then export PARADIS_DEV_PARTIAL=1 PARADIS_HIP_LIB=$R/build/variants/lib_advr4.so; else unset PARADIS_DEV_PARTIAL PARADIS_HIP_LIB; fi
  rm -rf $R/gpurun_out/r5/tr3_$lib
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5/tr3_$lib -o t -- python3 $R/tools/_adv_trace.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/r5/tr3_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:9]:
    if float(r["AverageNs"]) > 3000: print("%-72s calls %4s  avg %10.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
unset PARADIS_HIP_LIB PARADIS_DEV_PARTIAL
cd $R
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5/bench4.json 2> gpurun_out/r5/bench4.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench4.json'))
print('headline', d['value'], d['ms_per_step'])
for n,r in d['other_configs'].items():
    print(n, r.get('ms_per_step'), {k:(round(r[k]['frac'],3), round(r[k]['avg_launch_ms'],3)) for k in r if k.startswith('roofline')}, r.get('error'))
PY
