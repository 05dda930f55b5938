OUT=gpurun_out/r3_host
mkdir -p $OUT
for b in 1 4; do timeout -k 10 200 python bench.py --steps 10 --warmup 3 --batch $b --no-cpu-baseline --f32-steps 0 > $OUT/b$b.json 2> $OUT/b$b.err; echo "batch $b: $(grep 'timed' $OUT/b$b.err)"; done
