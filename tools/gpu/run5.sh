mkdir -p gpurun_out/r02e
python -m pytest tests -m gpu -q > gpurun_out/r02e/gputests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02e/gputests.log
tail -4 gpurun_out/r02e/gputests.log
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --shape-report gpurun_out/r02e/shapes.txt > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err
tail -2 gpurun_out/r02e/bench.err
