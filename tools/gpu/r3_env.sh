OUT=gpurun_out/r3_env
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_precision.py tests/test_gpu_kernels_isolated.py tests/test_gpu_nn.py -x -q > $OUT/tests.log 2>&1; rc=$?; tail -2 $OUT/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() { tag=$1; shift; timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag: $(grep 'timed' $OUT/$tag.err)"; }
run sp3072
run sp1024 --tune $((4<<20))
run sp2048 --tune $((8<<20))
run sp3840 --tune $((15<<20))
run sp3072b
