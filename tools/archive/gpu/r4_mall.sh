# r04: does processing FEWER cameras per pass (smaller tensors: producer -> consumer distance inside the 256 MiB Infinity Cache)
# and dropping the non-temporal hints pay?   bash tools/gpu/r4_mall.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_mall
mkdir -p $OUT
: > $OUT/ab.txt
for lib in head nont; do
  if [ $lib = head ]; then unset XAS_HIP_LIB; else export XAS_HIP_LIB=$GRAFT_REPO_ROOT/x-as-supervision_amd/xas_amd/abl/libxas_$lib.so; fi
  for cams in 8 4 2 1; do
    XAS_CAM_BATCH_MAX=$cams timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 --no-variant-check > $OUT/b.json 2> $OUT/b.err || { tail -3 $OUT/b.err; exit 1; }
    python3 -c "
import json; d=json.load(open('$OUT/b.json')); r=d['roofline']; print('$lib cams_per_pass=$cams', round(d['ms_per_step'],2), 'bn', round(r['batch_norm']['ms_per_step'],1), 'conv', round(r['conv_ms_per_step'],1))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
