#!/usr/bin/env python3
"""Step timeline from a rocprofv3 --kernel-trace database (rocpd sqlite): per phase of the optimisation step
(delimited by the two fused-Adam launches) wall time, per-stream busy time, idle gaps and the kernel families
that fill it.   python tools/timeline.py trace_results.db [step_index_from_end]"""
import re
import sqlite3
import sys
from collections import defaultdict


def family(name):
    n = re.sub(r'^void ', '', name)
    n = re.sub(r'\(.*$', '', n)
    n = n.replace('xas::', '')
    if n.startswith('at::native') or n.startswith('void at::'):
        return 'aten'
    return n[:60]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def main():
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if sys.argv[1].endswith('.csv.gz'):            # compact table written by tools/gpu/slim_trace.py
        import csv
        import gzip
        with gzip.open(sys.argv[1], 'rt') as f:
            rd = csv.reader(f)
            next(rd)
            rows = [(r[0], r[1], int(r[2]), int(r[3])) for r in rd]
    else:
        db = sqlite3.connect(sys.argv[1])
        rows = db.execute('select name, stream_id, start, end from kernels order by start').fetchall()
    adam = [i for i, r in enumerate(rows) if 'adam' in r[0]]
    # a step = [after adam 2k-1] .. [adam 2k+1]; phase A (disc update) ends at the first adam of the pair
    last = len(adam) - 1 - 2 * back              # index of the 2nd adam of the chosen step (skip the trailing serial step)
    a0, a1, a2 = adam[last - 2], adam[last - 1], adam[last]
    phases = [('disc step (4 detector fwd, no grad)', a0 + 1, a1 + 1), ('gen step (8 det fwd+bwd, 4 physique)', a1 + 1, a2 + 1)]
    print('kernels in trace: %d, adam launches: %d' % (len(rows), len(adam)))
    for title, lo, hi in phases:
        seg = rows[lo:hi]
        t0, t1 = seg[0][2], max(r[3] for r in seg)
        print('\n== %s: %d launches, wall %.2f ms' % (title, len(seg), (t1 - t0) / 1e6))
        # split the generator step at the first backward kernel (first bn_bwd / dgrad after the loss kernels)
        cut = None
        for i, r in enumerate(seg):
            if 'bwd' in r[0] or 'wgrad' in r[0]:
                cut = i
                break
        parts = [('all', seg)] if cut is None else [('forward', seg[:cut]), ('backward', seg[cut:])]
        for pname, part in parts:
            p0, p1 = part[0][2], max(r[3] for r in part)
            streams = defaultdict(list)
            fam = defaultdict(lambda: [0, 0.0])
            for n, s, a, b in part:
                streams[s].append((a, b))
                f = fam[family(n)]
                f[0] += 1
                f[1] += (b - a) / 1e6
            busy_any = union([iv for v in streams.values() for iv in v])
            print('  -- %s: %d launches, wall %.2f ms, GPU busy (any stream) %.2f ms, idle %.2f ms' %
                  (pname, len(part), (p1 - p0) / 1e6, busy_any / 1e6, (p1 - p0 - busy_any) / 1e6))
            for s, iv in sorted(streams.items(), key=lambda kv: -union(kv[1])):
                print('     stream %s: %d launches, busy %.2f ms' % (s, len(iv), union(iv) / 1e6))
            for k, (c, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:22]:
                print('     %-62s %5d  %8.2f ms' % (k, c, ms))


if __name__ == '__main__':
    main()
