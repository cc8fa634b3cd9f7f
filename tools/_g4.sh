mkdir -p gpurun_out/r5
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for lib in r5 r4; do
  if [ $lib = r4 ]; then export PARADIS_DEV_PARTIAL=1 PARADIS_HIP_LIB=$R/build/variants/lib_advr4.so; fi
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5/tr_$lib -o t -- python3 $R/tools/_adv_trace.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/r5/tr_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:14]:
    print("%-70s calls %4s  avg %10.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
unset PARADIS_HIP_LIB PARADIS_DEV_PARTIAL
cd $R
timeout 300 python -m pytest tests/test_hip_pad_advect.py -q -x -k "departure_centred" 2>&1 | grep -E "^E |assert|passed|failed" | head -20
