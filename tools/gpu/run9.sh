mkdir -p gpurun_out/r02j
python -m pytest tests/test_gpu_kernels_isolated.py tests/test_gpu_nn.py tests/test_gpu_groups.py tests/test_gpu_model.py -m gpu -q -x > gpurun_out/r02j/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r02j/tests.log
tail -4 gpurun_out/r02j/tests.log
python tools/ab_step.py 0 4194304 > gpurun_out/r02j/ab.log 2>&1; tail -4 gpurun_out/r02j/ab.log
