# per-shape HBM traffic of the conv entry points: two counter passes over tools/bench_conv.py (each entry point once per shape)
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3_pmc_shapes}
N=${2:-256}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export XAS_ONCE=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py all 1 $N bf16x6 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py all 1 $N bf16x6 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/mfma -o m --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py all 1 $N bf16x6 > $OUT/mfma.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_shapes_report.py $OUT $N > $OUT/shapes_traffic.txt
find $OUT -name "*.csv" -delete
cat $OUT/shapes_traffic.txt
