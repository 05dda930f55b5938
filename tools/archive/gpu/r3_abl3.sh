# timing ablations of igemm_x6_kernel (tools/build_abl.py): each mask on a few layer shapes, forward and data gradient
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3_abl3}
MASKS=${2:-"0 1 2 3 4 32 8 16 63"}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export XAS_SHAPES=${XAS_SHAPES:-1,8,12,10,11,17}
: > $OUT/abl.txt
for m in $MASKS; do
  if [ $m = 0 ]; then unset XAS_HIP_LIB; else export XAS_HIP_LIB=$GRAFT_REPO_ROOT/x-as-supervision_amd/xas_amd/abl/libxas_abl$m.so; fi
  echo "== ABL=$m (1 no weight loads, 2 no activation loads, 4 no split+LDS stores, 32 stores without split, 8 no barriers, 16 no fragment reads, 64 split work doubled)" >> $OUT/abl.txt
  timeout -k 10 120 python3 tools/bench_conv.py fwd 10 256 ${XAS_ABL_PREC:-f16x3} 2>&1 | grep -v amdgpu.ids >> $OUT/abl.txt || exit 1
  timeout -k 10 120 python3 tools/bench_conv.py dgrad 10 256 ${XAS_ABL_PREC:-f16x3} 2>&1 | grep -v amdgpu.ids >> $OUT/abl.txt || exit 1
done
cat $OUT/abl.txt
