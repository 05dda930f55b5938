# in-box A/B of several environment settings against the default: bash tools/gpu/ab_envs.sh "A=1;B=0;C=2 D=1" [rounds]   (';' separates settings)
R=${2:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_envs
mkdir -p $OUT
: > $OUT/ab.txt
IFS=';' read -ra SETS <<< "$1"
for i in $(seq 1 $R); do
  timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 > $OUT/b.json 2> $OUT/b.err || exit 1
  python3 -c "
import json; d=json.load(open('$OUT/b.json')); print('base', round(d['ms_per_step'],2))" >> $OUT/ab.txt
  for E in "${SETS[@]}"; do
    env $E timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 > $OUT/b.json 2> $OUT/b.err || { echo "$E FAILED" >> $OUT/ab.txt; continue; }
    python3 -c "
import json; d=json.load(open('$OUT/b.json')); print('[$E]', round(d['ms_per_step'],2))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
