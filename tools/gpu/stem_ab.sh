cd $GRAFT_REPO_ROOT
for v in head tpb1 tpb4 tpb16; do
  if [ $v = head ]; then unset XAS_HIP_LIB; else export XAS_HIP_LIB=$GRAFT_REPO_ROOT/x-as-supervision_amd/xas_amd/abl/libxas_$v.so; fi
  echo "== $v"; XAS_SHAPES=21 timeout -k 10 120 python3 tools/bench_conv.py fwd 20 256 bf16x6 2>&1 | grep "fwd" | head -1
done
