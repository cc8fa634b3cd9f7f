#!/bin/bash
# Round-6 late experiments on ONE box: the bf16-cotangent tests, the bf16-mixed step and the no-window gather A/B
# (development library: make -C paradis_model_amd/csrc dev).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/r6
timeout 900 python3 -m pytest tests/test_hip_amp.py tests/test_hip_blocks.py tests/test_hip_compile.py -q -x -m gpu > gpurun_out/r6/late_tests.log 2>&1
tail -5 gpurun_out/r6/late_tests.log
for i in 1 2; do
  python3 bench.py --amp --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra-legs --no-other-configs 2>/dev/null \
    | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print('amp step', r['ms_per_step'], 'ms')"
done | tee gpurun_out/r6/late_amp.txt
rm -f /tmp/adv_ref.pt
for d in 0 4 8; do
  PARADIS_HIP_LIB=paradis_model_amd/libparadis_hip_dev.so PARADIS_ADVECT_DIRECT=$d timeout 600 python3 tools/advect_direct_ab.py /tmp/adv_ref.pt
done > gpurun_out/r6/advect_direct_ab.txt 2>&1
cat gpurun_out/r6/advect_direct_ab.txt
