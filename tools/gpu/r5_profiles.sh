# round-5 evidence: default bench line (driver command), rocprofv3 kernel stats + timeline of that command
OUT=gpurun_out/${1:-r5_prof}
mkdir -p $OUT
timeout -k 10 900 python bench.py --shape-report $OUT/shapes.txt > $OUT/bench_b32.json 2> $OUT/bench_b32.err; tail -3 $OUT/bench_b32.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/xas_prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --f32-steps 0 --cse-steps 0 --no-variant-check > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1; python3 $GRAFT_REPO_ROOT/tools/gpu/slim_trace.py /tmp/xas_prof/trace_results.db $GRAFT_REPO_ROOT/$OUT; python3 $GRAFT_REPO_ROOT/tools/timeline.py /tmp/xas_prof/trace_results.db > $GRAFT_REPO_ROOT/$OUT/step_timeline.txt 2>&1 )
ls -la $OUT
for w in MPI_Multi_SurS1 HM36_Multi_SurS2; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --f32-steps 0 --cse-steps 0 --no-variant-check > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w: $(grep timed $OUT/bench_$w.err)"; done
timeout -k 10 300 python bench.py --workload HM36_Multi_SynthS2 --batch 64 --no-cpu-baseline --f32-steps 0 --cse-steps 0 --no-variant-check > $OUT/bench_synth_b64.json 2> $OUT/bench_synth_b64.err; echo "synth b64: $(grep timed $OUT/bench_synth_b64.err)"
timeout -k 10 300 python bench.py --precision bf16x6 --no-cpu-baseline --f32-steps 0 --cse-steps 0 --no-variant-check > $OUT/bench_b32_bf16x6.json 2> $OUT/bench_b32_bf16x6.err; echo "bf16x6: $(grep timed $OUT/bench_b32_bf16x6.err)"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --f32-steps 0 --cse-steps 0 > $OUT/bench_gloo2_b16.json 2> $OUT/bench_gloo2_b16.err; echo "gloo2: $(grep timed $OUT/bench_gloo2_b16.err)"
