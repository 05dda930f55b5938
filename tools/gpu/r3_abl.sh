OUT=gpurun_out/r3_abl
mkdir -p $OUT
for t in 0 2 4 8 16 32 14 30 62; do
  echo "tune=$t (ABL=$((t>>1)))"
  XAS_TUNE=$t XAS_SHAPES=8,12,17 timeout -k 10 100 python tools/bench_conv.py fwd 10 128 bf16x6 2>&1 | grep -v amdgpu.ids | grep -v TOTAL
done > $OUT/abl.txt 2>&1
cat $OUT/abl.txt
