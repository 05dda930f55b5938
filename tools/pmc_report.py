#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes collected by tools/gpu/pmc.sh (one bench.py step per pass, separate passes):
per kernel  launches | time share | MFMA-busy % (counter based) | HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE).

  python tools/pmc_report.py gpurun_out/r02_pmc profiles/r02_pmc_summary.md profiles/r02_conv_traffic.json

MFMA busy = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE/8 * SIMDs), the gfx94x MfmaUtil formula rocprofv3 ships
(GRBM_GUI_ACTIVE is reported summed over the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back); calibrated against a
register-only v_mfma_f32_32x32x2_f32 loop (tools/micro/mfma_peak) collected with the same counters in the same call.
FETCH_SIZE is in KiB and counts half of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM): doubled here."""
import collections
import csv
import glob
import json
import os
import re
import sys

SIMDS = 256 * 4
CONV = ('igemm_kernel', 'igemm_buf_kernel', 'wgrad_kernel', 'wgrad_buf_kernel', 'stem_fwd_kernel', 'stem_wgrad_kernel', 'igemm_x6_kernel', 'wgrad_x6_kernel',
        'igemm_x6t_kernel', 'wgrad_x6t_kernel')


def short(name):
    n = re.sub(r'^void ', '', name)
    n = re.sub(r'\(.*$', '', n).replace('xas::', '')
    if n.startswith('at::native'):
        return 'aten::' + n.split('<')[0].split('::')[-1]
    return n[:64]


def load(dirname):
    """-> {kernel: {counter: sum}}, {kernel: launches}, {kernel: ns}"""
    files = glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True)
    vals = collections.defaultdict(lambda: collections.defaultdict(float))
    seen, dur = collections.defaultdict(set), collections.defaultdict(float)
    for path in files:
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r['Kernel_Name'])
                vals[k][r['Counter_Name']] += float(r['Counter_Value'])
                did = r['Dispatch_Id']
                if did not in seen[k]:
                    seen[k].add(did)
                    if r.get('End_Timestamp') and r.get('Start_Timestamp'):
                        dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    return vals, {k: len(v) for k, v in seen.items()}, dur


def main():
    root, out_md = sys.argv[1], sys.argv[2]
    out_json = sys.argv[3] if len(sys.argv) > 3 else None
    rnd = sys.argv[4] if len(sys.argv) > 4 else 'r03'
    precision = sys.argv[5] if len(sys.argv) > 5 else 'bf16x6'
    collected = sys.argv[6] if len(sys.argv) > 6 else ''
    mv, mn, md = load(os.path.join(root, 'mfma'))
    fv, fn_, _ = load(os.path.join(root, 'fetch'))
    wv, wn, _ = load(os.path.join(root, 'write'))
    pv, pn, pd = load(os.path.join(root, 'peak')) if os.path.isdir(os.path.join(root, 'peak')) else ({}, {}, {})

    def util(v):
        g = v.get('GRBM_GUI_ACTIVE', 0.0)
        return 100.0 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (g / 8.0 * SIMDS) if g else 0.0

    lines = ['# %s - counter-based MFMA utilisation and HBM traffic per kernel (tools/pmc_report.py), precision %s' % (rnd, precision), '',
             'Source: three `rocprofv3 --pmc ... --kernel-trace` passes of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --f32-steps 0`',
             '(3 steps incl. warm-up and the serial step, B = 32 x 4 cameras per pass), collected by `tools/gpu/pmc.sh`.',
             'MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the fraction of the cycles the kernel runs in which the',
             'matrix pipe executes (a bf16x6 kernel at 100 % would deliver 419.5 TFLOP/s fp32-equivalent AT THE 2.4 GHz peak clock; the',
             'clock under these kernels is 1.7-2.0 GHz, clk column = GRBM_GUI_ACTIVE / 8 / duration).',
             'Kernels are serialised under counter collection, so durations here are NOT the overlapped step timings.', '']
    if pv:
        lines.append('Calibration, register-only `v_mfma_f32_32x32x2_f32` loop (`tools/micro/mfma_peak`), same counters:')
        for k, v in pv.items():
            lines.append('* `%s`: MFMA busy %.1f %% (%d launches)' % (k, util(v), pn[k]))
        lines.append('')
    tot = sum(md.values()) or 1.0
    lines += ['| kernel | launches | time share | avg us | MFMA busy % | clk GHz | HBM MB/launch (2xFETCH+WRITE) |', '|---|---|---|---|---|---|---|']
    conv_bytes, conv_launch, conv_busy, conv_gui = 0.0, 0, 0.0, 0.0
    for k in sorted(md, key=lambda k: -md[k])[:40]:
        n = mn[k]
        fb = 2 * fv.get(k, {}).get('FETCH_SIZE', 0.0) * 1024
        wb = wv.get(k, {}).get('WRITE_SIZE', 0.0) * 1024
        nn = max(1, fn_.get(k, n))
        lines.append('| `%s` | %d | %.1f %% | %.1f | %.1f | %.2f | %.1f |' % (
            k, n, 100 * md[k] / tot, md[k] / n / 1e3, util(mv[k]), mv[k].get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / max(1.0, md[k]), (fb + wb) / nn / 1e6))
    for k in md:
        if any(c in k for c in CONV):
            conv_bytes += 2 * fv.get(k, {}).get('FETCH_SIZE', 0.0) * 1024 + wv.get(k, {}).get('WRITE_SIZE', 0.0) * 1024
            conv_launch += fn_.get(k, mn[k])
            conv_busy += mv[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
            conv_gui += mv[k].get('GRBM_GUI_ACTIVE', 0.0)
    conv_util = 100.0 * conv_busy / (conv_gui / 8.0 * SIMDS) if conv_gui else 0.0
    lines += ['', 'Conv family (igemm* / wgrad* / stem): %d launches, **%.1f MB of HBM traffic per launch**, **MFMA busy %.1f %%** of '
              'the cycles the family runs.' % (conv_launch, conv_bytes / max(1, conv_launch) / 1e6, conv_util)]
    with open(out_md, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
    if out_json:
        with open(out_json, 'w') as f:
            json.dump({'what': 'HBM bytes per conv-family launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches from separate rocprofv3 '
                               '--pmc passes of bench.py --steps 1 --warmup 1 (camera-batched step, B=32 x 4 cameras); mfma_busy_pct = '
                               'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs) over the same launches',
                       'precision': precision, 'collected': collected,
                       'launches': conv_launch, 'bytes_per_launch': conv_bytes / max(1, conv_launch), 'mfma_busy_pct': conv_util}, f, indent=1)


if __name__ == '__main__':
    main()
