# micro-benchmark A/B of XAS_TUNE values on a shape list: bash tools/gpu/r3_tune_ab.sh <out> "<tune values>" [passes]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3_tune_ab}
TUNES=${2:-"0 2097152"}
PASSES=${3:-"fwd dgrad"}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export XAS_SHAPES=${XAS_SHAPES:-1,8,12,10,11,17,2,3,6,15}
: > $OUT/ab.txt
for r in 1 2; do
for t in $TUNES; do
  echo "== XAS_TUNE=$t" >> $OUT/ab.txt
  for ps in $PASSES; do
    XAS_TUNE=$t timeout -k 10 120 python3 tools/bench_conv.py $ps 10 256 bf16x6 2>&1 | grep -v amdgpu.ids >> $OUT/ab.txt || exit 1
  done
done
done
cat $OUT/ab.txt
