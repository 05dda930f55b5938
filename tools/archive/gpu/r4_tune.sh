# in-box A/B of xas_set_tuning values against 0:  bash tools/gpu/r4_tune.sh "32768 65536" [rounds]
R=${2:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_tune
mkdir -p $OUT
: > $OUT/ab.txt
for i in $(seq 1 $R); do
  for t in 0 $1; do
    timeout -k 10 300 python3 bench.py --tune $t --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 --no-variant-check > $OUT/b.json 2> $OUT/b.err || { tail -3 $OUT/b.err; exit 1; }
    python3 -c "
import json; d=json.load(open('$OUT/b.json')); r=d['roofline']; e=r['batch_norm']['by_entry']; print('tune=$t', round(d['ms_per_step'],2), 'bn', round(r['batch_norm']['ms_per_step'],1), {k.replace('xas_bn_',''): round(v['ms'],1) for k,v in e.items()})" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
