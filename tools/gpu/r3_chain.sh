OUT=gpurun_out/r3_chain
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_groups.py tests/test_gpu_parity_r3.py -x -q > $OUT/tests.log 2>&1; rc=$?; tail -5 $OUT/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag: $(grep 'timed' $OUT/$tag.err)"; }
run chains2 XAS_CHAINS=2
run chains1 XAS_CHAINS=1
run chains2b XAS_CHAINS=2
