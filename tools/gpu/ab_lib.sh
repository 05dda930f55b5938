# in-box A/B of two builds of the library on the benchmark step: bash tools/gpu/ab_lib.sh <other .so relative to the repo> [rounds]
OTHER=$GRAFT_REPO_ROOT/$1
R=${2:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_lib
mkdir -p $OUT
: > $OUT/ab.txt
for i in $(seq 1 $R); do
  for which in head other; do
    if [ $which = head ]; then unset XAS_HIP_LIB; else export XAS_HIP_LIB=$OTHER; fi
    timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --f32-steps 0 --cse-steps 0 > $OUT/b.json 2> $OUT/b.err || exit 1
    python3 -c "
import json; d=json.load(open('$OUT/b.json')); print('$which', round(d['ms_per_step'],2), 'serial x6 ms', round(d['roofline']['serial']['by_kernel_class']['bf16x6']['ms'],2))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
