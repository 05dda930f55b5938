#!/usr/bin/env python3
"""Per-shape HBM traffic and MFMA busy of the conv entry points from the counter passes of tools/gpu/pmc_shapes.sh
(tools/bench_conv.py with XAS_ONCE=1: per shape three normal-distribution fills, then the weight splits, forward, data
gradient, weight gradient + its slab reduction, each exactly once).  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB
(MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide coalesced stream on gfx950)."""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_conv_shapes import SHAPES, images_for

SIMDS = 1024


def dispatches(dirname):
    """-> list of (dispatch id, kernel name, {counter: value}, ns) in dispatch order"""
    rows = collections.OrderedDict()
    for path in glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                d = rows.setdefault(int(r['Dispatch_Id']), [r['Kernel_Name'], {}, 0.0])
                d[1][r['Counter_Name']] = d[1].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
                if r.get('End_Timestamp') and r.get('Start_Timestamp'):
                    d[2] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    return [(k,) + tuple(v) for k, v in sorted(rows.items())]


def per_shape(disp):
    """group the conv launches by shape: a run of normal-distribution fills starts a shape"""
    groups, cur, in_fill = [], None, False
    for _, name, ctr, ns in disp:
        fill = 'normal' in name or 'distribution' in name
        if fill and not in_fill:
            cur = collections.defaultdict(lambda: [collections.defaultdict(float), 0.0, ''])
            groups.append(cur)
        in_fill = fill
        if cur is None or fill:
            continue
        kind = None
        if 'igemm' in name or 'stem_fwd' in name or 'direct_fwd' in name or 'direct_dgrad' in name or 'thin' in name:
            kind = 'dgrad' if (', 1, ' in name or 'dgrad' in name) else 'fwd'
        if 'wgrad' in name or 'slab_reduce' in name:
            kind = 'wgrad'
        if kind is None:
            continue
        g = cur[kind]
        for c, v in ctr.items():
            g[0][c] += v
        g[1] += ns
        if 'slab' not in name:
            g[2] = name.split('(')[0].replace('void xas::', '')
    return groups


def main():
    root, n = sys.argv[1], int(sys.argv[2])
    f, w, m = (per_shape(dispatches(os.path.join(root, d))) for d in ('fetch', 'write', 'mfma'))
    print('# HBM traffic per conv launch by shape (N = %d images), bf16x6: measured MB | algorithmic MB | ratio | MFMA busy %% | us' % n)
    tot_m = tot_a = 0.0
    for i, shp in enumerate(SHAPES):
        if i >= len(f):
            break
        hi, wi, ci, co, r, st, pad = shp
        nn = images_for(n, shp)
        ho, wo = (hi + 2 * pad - r) // st + 1, (wi + 2 * pad - r) // st + 1
        xb, yb, wb = nn * hi * wi * ci * 4.0, nn * ho * wo * co * 4.0, co * r * r * ci * 4.0
        alg = {'fwd': xb + yb + 1.5 * wb, 'dgrad': xb + yb + 1.5 * wb, 'wgrad': xb + yb + wb}
        line = '%-30s' % str((nn,) + shp[:6])
        for kind in ('fwd', 'dgrad', 'wgrad'):
            fb = 2 * f[i][kind][0].get('FETCH_SIZE', 0.0) * 1024
            wbt = w[i][kind][0].get('WRITE_SIZE', 0.0) * 1024 if i < len(w) else 0.0
            mm = m[i][kind] if i < len(m) else [{}, 0.0, '']
            gui = mm[0].get('GRBM_GUI_ACTIVE', 0.0)
            busy = 100.0 * mm[0].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui / 8.0 * SIMDS) if gui else 0.0
            meas = fb + wbt
            tot_m += meas
            tot_a += alg[kind]
            line += ' | %s %7.0f %7.0f %4.2fx %4.1f%% %6.0fus %s' % (kind, meas / 1e6, alg[kind] / 1e6, meas / alg[kind], busy, mm[1] / 1e3,
                                                                  f[i][kind][2].replace('_kernel', '').replace(', false', ''))
        print(line)
    print('TOTAL measured %.1f GB, algorithmic %.1f GB, ratio %.2f' % (tot_m / 1e9, tot_a / 1e9, tot_m / max(1.0, tot_a)))


if __name__ == '__main__':
    main()
