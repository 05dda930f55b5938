# round-3 baseline: GPU suite at HEAD, per-shape conv numbers in both fp32-accurate modes, bench lines
OUT=gpurun_out/r3_base
mkdir -p $OUT
python -m pytest tests -m gpu -q -x > $OUT/gputests.log 2>&1; echo "tests rc=$?" >> $OUT/gputests.log
tail -5 $OUT/gputests.log
python tools/bench_conv.py all 10 > $OUT/conv_f32.txt 2>&1
XAS_PRECISION=2 python tools/bench_conv.py all 10 > $OUT/conv_x6.txt 2>&1
tail -4 $OUT/conv_f32.txt; tail -4 $OUT/conv_x6.txt
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --precision bf16x6 --shape-report $OUT/shapes_x6.txt > $OUT/bench_x6.json 2> $OUT/bench_x6.err
tail -2 $OUT/bench_x6.err
