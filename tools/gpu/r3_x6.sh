OUT=gpurun_out/r3_x6
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_precision.py tests/test_gpu_nn.py tests/test_gpu_kernels_isolated.py tests/test_gpu_parity_r3.py -x -q > $OUT/tests1.log 2>&1; rc=$?; echo "tests1 rc=$rc" >> $OUT/tests1.log
tail -8 $OUT/tests1.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/bench_conv.py all 10 128 bf16x6 > $OUT/conv_n128.txt 2>&1
grep -v amdgpu.ids $OUT/conv_n128.txt
