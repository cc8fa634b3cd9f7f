mkdir -p gpurun_out/r5
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_pad_advect.py -q -x 2>&1 | grep -E "^E |passed|failed" | head
timeout 900 python -m pytest tests/test_hip_amp.py -q -s 2>&1 | grep -E "MEASURED|^E |passed|failed|Error" | head -30
timeout 900 python -m pytest tests/test_hip_gemm_split.py -q -x 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5/tr2_r5 -o t -- python3 $R/tools/_adv_trace.py > /dev/null 2>&1
f=$(find $R/gpurun_out/r5/tr2_r5 -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:12]:
    print("%-70s calls %4s  avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cd $R
for lib in base wsync64 wsync32; do
  if [ $lib != base ]; then export PARADIS_HIP_LIB=$R/build/variants/lib_$lib.so; fi
  tools/wgrad_traffic.sh $lib 128 256 8
  tail -3 gpurun_out/wg_$lib.log
done
unset PARADIS_HIP_LIB
python tools/advect_bench.py 0.05 2>&1 | grep -v "generic\|amdgpu"
