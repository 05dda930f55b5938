# PMC passes over one benchmark step (separate passes, counters only with --kernel-trace: MI355X_MICROARCH.md / gpurun rules)
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03_pmc}
PREC=${2:-f16x3}
ROUND=${3:-r04}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --f32-steps 0 --cse-steps 0 --no-variant-check --precision $PREC"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d $OUT/mfma -o m --output-format csv -- $CMD > $OUT/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w --output-format csv -- $CMD > $OUT/write.log 2>&1
if [ -x $GRAFT_REPO_ROOT/tools/micro/mfma_peak ]; then
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d $OUT/peak -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/micro/mfma_peak > $OUT/peak.log 2>&1
fi
# summarise on the box; the raw counter CSVs (tens of MB) stay there
python3 $GRAFT_REPO_ROOT/tools/pmc_report.py $OUT $OUT/pmc_summary.md $OUT/conv_traffic.json $ROUND $PREC "$(date -u +%Y-%m-%d) tools/gpu/pmc.sh" > /dev/null
find $OUT -name "*.csv" -delete
ls -la $OUT
