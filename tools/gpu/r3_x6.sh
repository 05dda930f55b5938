# round 3: new bf16-split kernels - correctness first, then per-shape timings (camera-batched N)
OUT=gpurun_out/r3_x6
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_precision.py tests/test_gpu_nn.py tests/test_gpu_kernels_isolated.py tests/test_gpu_parity_r3.py -x -q > $OUT/tests1.log 2>&1; rc=$?; echo "tests1 rc=$rc" >> $OUT/tests1.log
tail -15 $OUT/tests1.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/bench_conv.py all 10 128 bf16x6,f32 > $OUT/conv_n128.txt 2>&1
tail -30 $OUT/conv_n128.txt
