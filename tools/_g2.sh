mkdir -p gpurun_out/r5
python -m pytest tests/test_hip_pad_advect.py -q -x > gpurun_out/r5/t_adv.log 2>&1
tail -5 gpurun_out/r5/t_adv.log
python tools/advect_bench.py 0.05 > gpurun_out/r5/advbench_shift.log 2>&1
grep -v generic gpurun_out/r5/advbench_shift.log
python -m pytest tests/test_hip_determinism.py tests/test_hip_configs.py -q -x -k "not gradients_fp64 and not one_layer" > gpurun_out/r5/t_cfg.log 2>&1
tail -5 gpurun_out/r5/t_cfg.log
