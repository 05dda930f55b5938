# in-box A/B of SEVERAL builds of the library on the benchmark step, interleaved: bash tools/gpu/ab_libs.sh <rounds> <name> [<name> ...]
# (names of x-as-supervision_amd/xas_amd/abl/libxas_<name>.so, built by tools/build_variant.py; "head" = the shipped library)
R=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_libs
mkdir -p $OUT
: > $OUT/ab.txt
for i in $(seq 1 $R); do
  for which in head "$@"; do
    if [ $which = head ]; then unset XAS_HIP_LIB; else export XAS_HIP_LIB=$GRAFT_REPO_ROOT/x-as-supervision_amd/xas_amd/abl/libxas_$which.so; fi
    timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --f32-steps 0 --cse-steps 0 --no-variant-check 2> $OUT/b.err > /dev/null || exit 1
    echo "$which $(grep -a timed $OUT/b.err)" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
