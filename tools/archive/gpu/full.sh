# full GPU check: tests, bench (default command), optional trace
OUT=gpurun_out/${1:-full}
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; echo "tests rc=$?" >> $OUT/gputests.log
tail -4 $OUT/gputests.log
python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err
