set -x
mkdir -p gpurun_out/r02b
python -m pytest tests/test_gpu_kernels_isolated.py tests/test_gpu_model.py -m gpu -q -x > gpurun_out/r02b/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r02b/tests.log
python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --batch 4 > gpurun_out/r02b/bench2.json 2> gpurun_out/r02b/bench2.err; echo "rc=$?" >> gpurun_out/r02b/bench2.err
