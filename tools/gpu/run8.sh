mkdir -p gpurun_out/r02i
python -m pytest tests/test_gpu_kernels_isolated.py tests/test_gpu_nn.py tests/test_gpu_groups.py -m gpu -q -x > gpurun_out/r02i/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r02i/tests.log
tail -12 gpurun_out/r02i/tests.log
python tools/ab_step.py 0 > gpurun_out/r02i/ab.log 2>&1; tail -3 gpurun_out/r02i/ab.log
XAS_BN_MASK=0 python tools/ab_step.py 0 > gpurun_out/r02i/ab0.log 2>&1; tail -3 gpurun_out/r02i/ab0.log
