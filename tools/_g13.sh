mkdir -p gpurun_out/r5
tools/ab_libs.sh "shipped gemmbase" 3 > gpurun_out/r5/ab_gemm.txt 2>&1; cat gpurun_out/r5/ab_gemm.txt
python -m pytest tests -m gpu -q -x > gpurun_out/r5/t_full.log 2>&1; tail -4 gpurun_out/r5/t_full.log
