# in-box A/B of several library builds against HEAD: bash tools/gpu/ab_libs.sh "name1 name2 ..." [rounds]
R=${2:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_libs
mkdir -p $OUT
: > $OUT/ab.txt
for i in $(seq 1 $R); do
  for which in head $1; do
    if [ $which = head ]; then unset XAS_HIP_LIB; else export XAS_HIP_LIB=$GRAFT_REPO_ROOT/x-as-supervision_amd/xas_amd/abl/libxas_$which.so; fi
    timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 > $OUT/b.json 2> $OUT/b.err || exit 1
    python3 -c "
import json; d=json.load(open('$OUT/b.json')); print('$which', round(d['ms_per_step'],2))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
