# standalone weight-gradient rates, head (depth 3 where it fits) vs depth-2 build; then the step
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_wgrad
mkdir -p $OUT
for lib in head depth2; do
  if [ $lib = head ]; then unset XAS_HIP_LIB; else export XAS_HIP_LIB=$GRAFT_REPO_ROOT/x-as-supervision_amd/xas_amd/abl/libxas_$lib.so; fi
  python3 tools/bench_conv.py wgrad 20 256 f16x3 > $OUT/wgrad_$lib.txt 2>&1
done
unset XAS_HIP_LIB
paste -d'|' <(cut -c1-80 $OUT/wgrad_head.txt) <(cut -c36-80 $OUT/wgrad_depth2.txt)
bash tools/gpu/ab_libs.sh "depth2" 3 2>&1 | tail -6
