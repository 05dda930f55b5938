# standalone weight-gradient rates of library variants (bash tools/gpu/r4_wgrad.sh "name1 name2") on the shapes XAS_SHAPES selects
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_wgrad
mkdir -p $OUT
for lib in head $1; do
  if [ $lib = head ]; then unset XAS_HIP_LIB; else export XAS_HIP_LIB=$GRAFT_REPO_ROOT/x-as-supervision_amd/xas_amd/abl/libxas_$lib.so; fi
  python3 tools/bench_conv.py wgrad 20 256 ${PRECS:-f16x3} 2>&1 | grep -v libdrm > $OUT/wgrad_$lib.txt
  echo "== $lib"; cut -c1-130 $OUT/wgrad_$lib.txt
done
