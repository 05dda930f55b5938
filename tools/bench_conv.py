#!/usr/bin/env python3
"""Micro-benchmark of the conv entry points on the detector's layer shapes (B=32).
usage: python tools/bench_conv.py [fwd|dgrad|wgrad|all] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch

from xas_amd import _lib
from xas_amd._lib import ConvShape, call, ptr, query

SHAPES = [  # N, Hi, Wi, Cin, Cout, R, stride, pad
    (32, 64, 64, 64, 64, 1, 1, 0), (32, 64, 64, 64, 64, 3, 1, 1), (32, 64, 64, 64, 256, 1, 1, 0),
    (32, 64, 64, 256, 64, 1, 1, 0), (32, 64, 64, 256, 128, 1, 1, 0), (32, 64, 64, 128, 128, 3, 2, 1),
    (32, 32, 32, 128, 512, 1, 1, 0), (32, 32, 32, 512, 128, 1, 1, 0), (32, 32, 32, 128, 128, 3, 1, 1),
    (32, 32, 32, 256, 256, 3, 2, 1), (32, 16, 16, 256, 1024, 1, 1, 0), (32, 16, 16, 1024, 256, 1, 1, 0),
    (32, 16, 16, 256, 256, 3, 1, 1), (32, 16, 16, 512, 512, 3, 2, 1), (32, 8, 8, 512, 2048, 1, 1, 0),
    (32, 8, 8, 2048, 512, 1, 1, 0), (32, 8, 8, 512, 512, 3, 1, 1), (32, 64, 64, 256, 1152, 1, 1, 0),
    (32, 128, 128, 64, 64, 3, 1, 1), (32, 256, 256, 32, 32, 3, 1, 1), (32, 256, 256, 64, 32, 3, 1, 1),
]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device('cuda')
    query('xas_set_tuning', int(os.environ.get('XAS_TUNE', '0')))
    tot = {}
    for (n, hi, wi, ci, co, r, st, pad) in SHAPES:
        ho, wo = (hi + 2 * pad - r) // st + 1, (wi + 2 * pad - r) // st + 1
        shp = ConvShape(n, hi, wi, ci, co, r, r, st, pad, ho, wo)
        x = torch.randn(n * hi * wi * ci, device=dev)
        dy = torch.randn(n * ho * wo * co, device=dev)
        w = torch.randn(co * r * r * ci, device=dev) * 0.05
        y = torch.empty_like(dy)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp)), device=dev)
        flops = 2.0 * n * ho * wo * co * r * r * ci
        runs = {'fwd': lambda: call('xas_conv_fwd', ptr(x), ptr(w), None, ptr(y), shp),
                'dgrad': lambda: call('xas_conv_dgrad', ptr(dy), ptr(w), ptr(dx), shp),
                'wgrad': lambda: call('xas_conv_wgrad_oihw', ptr(x), ptr(dy), ptr(dw), ptr(ws), shp)}
        line = '%-34s' % str((n, hi, wi, ci, co, r, st))
        for k, fn in runs.items():
            if which not in ('all', k):
                continue
            fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
            line += '  %s %7.1f us %6.1f TF' % (k, ms * 1e3, flops / ms / 1e9)
            t = tot.setdefault(k, [0.0, 0.0])
            t[0] += flops
            t[1] += ms
        print(line, flush=True)
    for k, (f, ms) in tot.items():
        print('TOTAL %-6s %.2f ms  %.1f TF' % (k, ms, f / ms / 1e9))


if __name__ == '__main__':
    main()
