# r04: what does folding the batch-norm backward reduction into the data-gradient epilogue buy?  (exists for bf16x6 only)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_dgradbn
mkdir -p $OUT
: > $OUT/ab.txt
for i in 1 2; do
  for f in 0 1; do
    XAS_DGRAD_BN=$f timeout -k 10 300 python3 bench.py --precision bf16x6 --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 --no-variant-check > $OUT/b.json 2> $OUT/b.err || { tail -3 $OUT/b.err; exit 1; }
    python3 -c "
import json; d=json.load(open('$OUT/b.json')); r=d['roofline']; e=r['batch_norm']['by_entry']; print('bf16x6 XAS_DGRAD_BN=$f', round(d['ms_per_step'],2), 'bn', round(r['batch_norm']['ms_per_step'],1), 'conv', round(r['conv_ms_per_step'],1), {k.replace('xas_bn_',''): round(v['ms'],1) for k,v in e.items()})" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
