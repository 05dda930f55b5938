# round-5 evidence: default bench line (driver command), rocprofv3 kernel stats + timeline of that command
OUT=gpurun_out/${1:-r5_prof}
mkdir -p $OUT
timeout -k 10 900 python bench.py --shape-report $OUT/shapes.txt > $OUT/bench_b32.json 2> $OUT/bench_b32.err; tail -3 $OUT/bench_b32.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/xas_prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --f32-steps 0 --no-variant-check > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1; python3 $GRAFT_REPO_ROOT/tools/gpu/slim_trace.py /tmp/xas_prof/trace_results.db $GRAFT_REPO_ROOT/$OUT; python3 $GRAFT_REPO_ROOT/tools/timeline.py /tmp/xas_prof/trace_results.db > $GRAFT_REPO_ROOT/$OUT/step_timeline.txt 2>&1 )
ls -la $OUT
