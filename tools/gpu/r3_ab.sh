OUT=gpurun_out/r3_ab
mkdir -p $OUT
XAS_TUNE=0 XAS_SHAPES=1,4,8,17,18 timeout -k 10 200 python tools/bench_conv.py fwd 10 128 bf16x6,f32 > $OUT/t0.txt 2>&1
XAS_TUNE=1 XAS_SHAPES=1,4,8,17,18 timeout -k 10 200 python tools/bench_conv.py fwd 10 128 bf16x6,f32 > $OUT/t1.txt 2>&1
paste -d'\n' $OUT/t0.txt $OUT/t1.txt | grep -v amdgpu.ids
