# in-box A/B of environment settings on the benchmark step: bash tools/gpu/ab_env.sh "<VAR=val ...>" [rounds]
SET="$1"
R=${2:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_env
mkdir -p $OUT
: > $OUT/ab.txt
for i in $(seq 1 $R); do
  for which in base other; do
    if [ $which = base ]; then E=""; else E="$SET"; fi
    env $E timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 --cse-steps 0 > $OUT/b.json 2> $OUT/b.err || exit 1
    python3 -c "
import json; d=json.load(open('$OUT/b.json')); print('$which [$E]', round(d['ms_per_step'],2))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
