mkdir -p gpurun_out/r02k
python -m pytest tests/test_gpu_bf16.py -m gpu -q -x -s > gpurun_out/r02k/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r02k/tests.log
tail -8 gpurun_out/r02k/tests.log | cut -c1-220
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --precision bf16 --shape-report gpurun_out/r02k/shapes_bf16.txt > gpurun_out/r02k/bench_bf16.json 2> gpurun_out/r02k/bench_bf16.err; tail -2 gpurun_out/r02k/bench_bf16.err
