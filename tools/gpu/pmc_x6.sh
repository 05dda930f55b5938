# counters of the bf16-split conv kernels on a few shapes (separate passes; --pmc only with --kernel-trace)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3_pmc_x6}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/tools/x6_probe.py 3 bf16x6"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/a -o a --output-format csv -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace -d $OUT/b -o b --output-format csv -- $CMD > $OUT/b.log 2>&1
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, collections, os, re
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r3_pmc_x6'
for sub in ('a', 'b'):
    vals = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); seen = collections.defaultdict(set)
    for path in glob.glob(out + '/' + sub + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            k = re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void xas::', '')
            if 'x6' not in k and 'igemm' not in k and 'wgrad' not in k: continue
            k = k + ' grid=' + r.get('Grid_Size', '?')
            vals[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Dispatch_Id'] not in seen[k]:
                seen[k].add(r['Dispatch_Id']); dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    for k in sorted(vals):
        v = vals[k]; n = len(seen[k])
        line = '%-70s n=%d avg_us=%.1f ' % (k[:70], n, dur[k] / n / 1e3)
        if 'GRBM_GUI_ACTIVE' in v:
            g = v['GRBM_GUI_ACTIVE'] / 8.0
            line += 'clk=%.2fGHz mfma_busy=%.1f%% ' % (g / dur[k], 100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (g * 1024))
            wc = v['SQ_WAVE_CYCLES']
            line += 'wait_any=%.1f%% wait_inst=%.1f%% valu=%.1f%% lds=%.1f%% ' % (100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc, 100 * v['SQ_ACTIVE_INST_VALU'] / wc, 100 * v['SQ_ACTIVE_INST_LDS'] / wc)
        else:
            line += 'lds_conflict/idx=%.2f insts valu=%.3g lds=%.3g vmem=%.3g salu=%.3g wait_lds=%.3g' % (v['SQ_LDS_BANK_CONFLICT'] / max(1, v['SQ_LDS_IDX_ACTIVE']), v['SQ_INSTS_VALU'] / n, v['SQ_INSTS_LDS'] / n, v['SQ_INSTS_VMEM_RD'] / n, v['SQ_INSTS_SALU'] / n, v['SQ_WAIT_INST_LDS'] / n)
        print(line)
PY
find $OUT -name "*.csv" -delete
cat $OUT/summary.txt
