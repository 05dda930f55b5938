# round-3 evidence: full GPU suite, default bench line (driver command), rocprofv3 kernel stats of that command, other configs
OUT=gpurun_out/${1:-r3_prof}
mkdir -p $OUT
timeout -k 10 1500 python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; echo "tests rc=$?" >> $OUT/gputests.log
tail -3 $OUT/gputests.log
timeout -k 10 600 python bench.py --shape-report $OUT/shapes.txt > $OUT/bench_b32.json 2> $OUT/bench_b32.err; tail -3 $OUT/bench_b32.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/xas_prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --f32-steps 0 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1; python3 $GRAFT_REPO_ROOT/tools/gpu/slim_trace.py /tmp/xas_prof/trace_results.db $GRAFT_REPO_ROOT/$OUT; python3 $GRAFT_REPO_ROOT/tools/timeline.py /tmp/xas_prof/trace_results.db > $GRAFT_REPO_ROOT/$OUT/step_timeline.txt 2>&1 )
for w in MPI_Multi_SurS1 HM36_Multi_SurS2; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --f32-steps 0 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w: $(grep timed $OUT/bench_$w.err)"; done
timeout -k 10 300 python bench.py --workload HM36_Multi_SynthS2 --batch 64 --no-cpu-baseline --f32-steps 0 > $OUT/bench_synth_b64.json 2> $OUT/bench_synth_b64.err; echo "synth b64: $(grep timed $OUT/bench_synth_b64.err)"
timeout -k 10 300 python bench.py --precision f32 --no-cpu-baseline > $OUT/bench_b32_f32.json 2> $OUT/bench_b32_f32.err; echo "f32: $(grep timed $OUT/bench_b32_f32.err)"
timeout -k 10 300 python bench.py --precision bf16x6 --no-cpu-baseline --f32-steps 0 > $OUT/bench_b32_bf16x6.json 2> $OUT/bench_b32_bf16x6.err; echo "bf16x6: $(grep timed $OUT/bench_b32_bf16x6.err)"
timeout -k 10 300 python bench.py --precision bf16 --no-cpu-baseline --f32-steps 0 > $OUT/bench_b32_bf16.json 2> $OUT/bench_b32_bf16.err; echo "bf16: $(grep timed $OUT/bench_b32_bf16.err)"
