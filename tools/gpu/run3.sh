mkdir -p gpurun_out/r02c
python -m pytest tests/test_gpu_groups.py -m gpu -q -x > gpurun_out/r02c/groups.log 2>&1; echo "rc=$?" >> gpurun_out/r02c/groups.log
tail -30 gpurun_out/r02c/groups.log
