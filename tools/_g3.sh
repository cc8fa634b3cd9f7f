mkdir -p gpurun_out/r5
export ADVECT_BENCH_ONLY=721x1440
for i in 1 2; do
PARADIS_DEV_PARTIAL=1 PARADIS_HIP_LIB=build/variants/lib_advr4.so python tools/advect_bench.py 0.05 2>&1 | grep -v "amdgpu.ids" | sed 's/^/r4  /'
python tools/advect_bench.py 0.05 2>&1 | grep -v "amdgpu.ids" | sed 's/^/r5  /'
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5/adv721 -o t -- python3 $GRAFT_REPO_ROOT/tools/advect_bench.py 0.05 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r5/adv721 -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
python -m pytest tests/test_hip_pad_advect.py -q -x -k "departure_centred" 2>&1 | tail -3
