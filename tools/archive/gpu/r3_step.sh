# quick loop: conv tests, short bench, kernel trace of 2 steps
OUT=gpurun_out/${1:-r3_step}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_precision.py tests/test_gpu_parity_r3.py tests/test_gpu_kernels_isolated.py -x -q > $OUT/tests.log 2>&1; rc=$?; tail -3 $OUT/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 --shape-report $OUT/shapes.txt > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
bash tools/gpu/trace.sh ${1:-r3_step}/trace > /dev/null 2>&1
python3 - <<'PY'
import csv,sys,os
p=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/'+(sys.argv[1] if len(sys.argv)>1 else 'r3_step')+'/trace/kernel_stats.csv'
PY
