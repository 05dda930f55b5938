# full GPU check: the whole -m gpu suite, then the default bench line with shape reports
OUT=gpurun_out/${1:-r3_full}
mkdir -p $OUT
timeout -k 10 1500 python -m pytest tests -m gpu -q -x > $OUT/gputests.log 2>&1; echo "tests rc=$?" >> $OUT/gputests.log
tail -6 $OUT/gputests.log
timeout -k 10 600 python bench.py --steps 8 --warmup 3 --shape-report $OUT/shapes.txt > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err
