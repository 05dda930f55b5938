#!/usr/bin/env python3
"""Per-kernel roofline table from three rocprofv3 runs of the same command (bench.py --steps 1 --warmup 1):
    --kernel-trace --stats            -> <stats>.csv        (durations)
    --pmc FETCH_SIZE --kernel-trace   -> <fetch>_counter_collection.csv
    --pmc WRITE_SIZE --kernel-trace   -> <write>_counter_collection.csv
usage: python tools/kernel_roofline.py stats.csv fetch_counter_collection.csv write_counter_collection.csv out.md
HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md).  Durations come from the
stats run, traffic from the counter runs (same launches, different processes)."""
import collections
import csv
import sys


def key(name):
    return name.split('(')[0].replace('void ', '').strip()


def counters(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get('Counter_Name') == counter:
                k = key(r['Kernel_Name'])
                per[k][0] += 1
                per[k][1] += float(r['Counter_Value'])
    return per


def main():
    stats, fpath, wpath, out = sys.argv[1:5]
    dur = {}
    with open(stats) as f:
        for r in csv.DictReader(f):
            dur[key(r['Name'])] = (int(r['Calls']), int(r['TotalDurationNs']))
    fe, wr = counters(fpath, 'FETCH_SIZE'), counters(wpath, 'WRITE_SIZE')
    rows = []
    for k, (calls, ns) in dur.items():
        if k not in fe:
            continue
        gib = (2 * fe[k][1] + wr.get(k, [0, 0.0])[1]) / 2 ** 20
        rows.append((ns, k, calls, gib, gib * 2 ** 30 / (ns * 1e-9) / 1e12))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    with open(out, 'w') as f:
        f.write('| kernel | launches | time share | avg us | HBM GiB (2*FETCH+WRITE) | HBM TB/s while running |\n|---|---|---|---|---|---|\n')
        for ns, k, calls, gib, tbs in rows[:28]:
            f.write('| `%s` | %d | %.1f %% | %.1f | %.1f | %.2f |\n' % (k[:70], calls, 100.0 * ns / tot, ns / calls / 1e3, gib, tbs))
    print(open(out).read())


if __name__ == '__main__':
    main()
