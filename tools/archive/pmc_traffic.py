#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same bench.py command into HBM bytes per launch of
the conv kernel family (what bench.py reports as roofline.traffic).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv profiles/r01_conv_traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced stream
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), hence the factor 2."""
import collections
import csv
import json
import sys

FAMILY = ('igemm_kernel', 'igemm_buf_kernel', 'wgrad_kernel', 'wgrad_buf_kernel', 'stem_fwd_kernel')


def load(path, counter):
    tot, n, per = 0.0, 0, collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get('Counter_Name') != counter:
                continue
            name = r['Kernel_Name']
            v = float(r['Counter_Value'])
            key = name.split('(')[0].replace('void ', '')
            per[key][0] += 1
            per[key][1] += v
            if any(k in name for k in FAMILY):
                tot += v
                n += 1
    return tot, n, per


def main():
    fpath, wpath, out = sys.argv[1:4]
    f_kib, nf, perf = load(fpath, 'FETCH_SIZE')
    w_kib, nw, perw = load(wpath, 'WRITE_SIZE')
    assert nf == nw and nf > 0, (nf, nw)
    res = {'what': 'HBM bytes per conv-family launch (igemm*/wgrad*/stem kernels) = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches, '
                   'from two rocprofv3 --pmc passes of bench.py --steps 1 --warmup 1 (B=32); FETCH_SIZE doubled per '
                   'MI355X_MICROARCH.md HBM note',
           'launches': nf, 'fetch_kib': f_kib, 'write_kib': w_kib,
           'bytes_per_launch': (2 * f_kib + w_kib) * 1024 / nf}
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))
    print('\nper kernel family: launches, GiB fetched (x2), GiB written')
    for k in sorted(perf, key=lambda k: -perf[k][1])[:25]:
        print('%-60s %6d %9.2f %9.2f' % (k[:60], perf[k][0], 2 * perf[k][1] / 2 ** 20, perw.get(k, [0, 0.0])[1] / 2 ** 20))


if __name__ == '__main__':
    main()
