# DMA-staged igemm (default) vs register-staged (tune bit 20): parity tests, then the micro-benchmark on a few shapes
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3_v4}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_precision.py tests/test_gpu_parity_r3.py tests/test_gpu_nn.py -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?; tail -3 $OUT/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
export XAS_SHAPES=${XAS_SHAPES:-1,8,12,10,11,17,2,3,6,15}
: > $OUT/ab.txt
for t in 0 1048576 0 1048576; do
  echo "== XAS_TUNE=$t" >> $OUT/ab.txt
  XAS_TUNE=$t timeout -k 10 120 python3 tools/bench_conv.py fwd 10 256 bf16x6 2>&1 | grep -v amdgpu.ids >> $OUT/ab.txt || exit 1
  XAS_TUNE=$t timeout -k 10 120 python3 tools/bench_conv.py dgrad 10 256 bf16x6 2>&1 | grep -v amdgpu.ids >> $OUT/ab.txt || exit 1
done
cat $OUT/ab.txt
