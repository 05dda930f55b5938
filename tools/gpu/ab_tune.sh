# in-box A/B of bench.py --tune values on the benchmark step: bash tools/gpu/ab_tune.sh "<tune values>" [rounds]
R=${2:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_tune
mkdir -p $OUT
: > $OUT/ab.txt
for i in $(seq 1 $R); do
  for t in $1; do
    timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 --cse-steps 0 --tune $t > $OUT/b.json 2> $OUT/b.err || exit 1
    python3 -c "
import json; d=json.load(open('$OUT/b.json')); print('tune $t', round(d['ms_per_step'],2), 'serial conv', round(d['roofline']['serial']['by_kernel_class'].get('f16x3', d['roofline']['serial']['by_kernel_class'].get('bf16x6'))['ms'],2))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
