OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-trace}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/xas_prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --f32-steps 0 --cse-steps 0 --no-variant-check > $OUT/prof.log 2>&1
# the rocpd database is ~40 MB: keep a compact table (name, stream, start, end) and the top-kernel statistics only
python3 $GRAFT_REPO_ROOT/tools/gpu/slim_trace.py /tmp/xas_prof/trace_results.db $OUT
ls -la $OUT
