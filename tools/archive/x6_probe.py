#!/usr/bin/env python3
"""A handful of conv launches for counter collection (rocprofv3 --pmc): python tools/x6_probe.py [reps] [prec]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd'), os.path.join(ROOT, 'tools')]
import torch

from bench_conv import weights_for
from xas_amd import _lib
from xas_amd._lib import ConvShape, call, ptr, query

CASES = [(128, 32, 32, 128, 128, 3, 1, 1), (112, 64, 64, 256, 1152, 1, 1, 0), (128, 64, 64, 64, 256, 1, 1, 0),
         (128, 16, 16, 256, 256, 3, 1, 1), (128, 128, 128, 64, 64, 3, 1, 1)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    prec = sys.argv[2] if len(sys.argv) > 2 else 'bf16x6'
    dev = torch.device('cuda')
    for (n, hi, wi, ci, co, r, st, pad) in CASES:
        ho, wo = (hi + 2 * pad - r) // st + 1, (wi + 2 * pad - r) // st + 1
        shp = ConvShape(n, hi, wi, ci, co, r, r, st, pad, ho, wo, 1 + _lib.PREC_NAMES[prec])
        x = torch.randn(n * hi * wi * ci, device=dev)
        dy = torch.randn(n * ho * wo * co, device=dev)
        w = torch.randn(co * r * r * ci, device=dev) * 0.05
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)
        ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp)), device=dev)
        wf, wt = weights_for(shp, w, 0), weights_for(shp, w, 1)
        for _ in range(reps):
            call('xas_conv_fwd', ptr(x), ptr(wf), None, ptr(y), shp)
            call('xas_conv_dgrad', ptr(dy), ptr(wt), ptr(dx), shp)
            call('xas_conv_wgrad_oihw', ptr(x), ptr(dy), ptr(dw), ptr(ws), shp)
        torch.cuda.synchronize()


if __name__ == '__main__':
    main()
