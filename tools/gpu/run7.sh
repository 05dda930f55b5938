mkdir -p gpurun_out/r02f
python -m pytest tests/test_gpu_nn.py tests/test_gpu_kernels_isolated.py tests/test_gpu_groups.py -m gpu -q -x > gpurun_out/r02f/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02f/tests.log
tail -3 gpurun_out/r02f/tests.log
python tools/ab_step.py 0 1048576 2097152 > gpurun_out/r02f/ab.log 2>&1; tail -5 gpurun_out/r02f/ab.log
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --shape-report gpurun_out/r02f/shapes.txt > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err
tail -1 gpurun_out/r02f/bench.err
