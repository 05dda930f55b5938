OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-trace_x6}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/xas_prof6 -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision bf16x6 > $OUT/prof.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/gpu/slim_trace.py /tmp/xas_prof6/trace_results.db $OUT
ls -la $OUT
