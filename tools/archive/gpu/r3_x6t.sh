OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3_x6t}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 5 200 python -m pytest tests/test_gpu_precision.py -x -q -m gpu -k "bf16x6_is_fp32_accurate or every_layer" > $OUT/tests0.log 2>&1; rc=$?; tail -3 $OUT/tests0.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python -m pytest tests/test_gpu_precision.py tests/test_gpu_parity_r3.py tests/test_gpu_nn.py tests/test_gpu_groups.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; tail -3 $OUT/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
XAS_SHAPES=1,8,12,16,18,19,20 bash tools/gpu/r3_tune_ab.sh ${1:-r3_x6t}/ab "0 4194304"
