# like ab_libs.sh with extra bench arguments: bash tools/gpu/ab_libs_b.sh "names" rounds "extra bench args"
R=${2:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_libs
mkdir -p $OUT
: > $OUT/ab.txt
for i in $(seq 1 $R); do
  for which in head $1; do
    if [ $which = head ]; then unset XAS_HIP_LIB; else export XAS_HIP_LIB=$GRAFT_REPO_ROOT/x-as-supervision_amd/xas_amd/abl/libxas_$which.so; fi
    timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --f32-steps 0 $3 > $OUT/b.json 2> $OUT/b.err || exit 1
    python3 -c "
import json; d=json.load(open('$OUT/b.json')); print('$which', round(d['ms_per_step'],2))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
