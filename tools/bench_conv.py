#!/usr/bin/env python3
"""Micro-benchmark (and quick cross-check) of the conv entry points on the detector's and the physique net's layer shapes.
usage: python tools/bench_conv.py [fwd|dgrad|wgrad|all] [reps] [N] [precisions, e.g. bf16x6,f32]
Each precision mode is timed on the same tensors; the bf16x6 results are compared with the exact-fp32 MFMA results."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch

from xas_amd import _lib
from xas_amd._lib import ConvShape, call, ptr, query
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_conv_shapes import SHAPES, images_for



def weights_for(shp, w, transposed):
    """weight buffer in the format the library wants for this shape / precision (xas_conv_weight_planes)"""
    planes = query('xas_conv_weight_planes', shp, transposed)
    if not planes:
        return w
    rows = shp.Cin if transposed else shp.Cout
    kk = w.numel() // rows
    sp = torch.empty(query('xas_split_weight_bytes', rows, kk, planes), device=w.device, dtype=torch.uint8)
    call('xas_split_weight', ptr(w), ptr(sp), rows, kk, planes)
    return sp


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    precs = (sys.argv[4] if len(sys.argv) > 4 else 'bf16x6,f32').split(',')
    dev = torch.device('cuda')
    query('xas_set_tuning', int(os.environ.get('XAS_TUNE', '0')))
    tot = {}
    once = os.environ.get('XAS_ONCE') == '1'      # counter passes: every entry point exactly once per shape
    sel = os.environ.get('XAS_SHAPES')
    shapes = [SHAPES[int(i)] for i in sel.split(',')] if sel else SHAPES
    for (hi, wi, ci, co, r, st, pad) in shapes:
        nn = images_for(n, (hi, wi, ci, co, r, st, pad))
        ho, wo = (hi + 2 * pad - r) // st + 1, (wi + 2 * pad - r) // st + 1
        x = torch.randn(nn * hi * wi * ci, device=dev)
        dy = torch.randn(nn * ho * wo * co, device=dev)
        w = torch.randn(co * r * r * ci, device=dev) * 0.05
        flops = 2.0 * nn * ho * wo * co * r * r * ci
        line = '%-34s' % str((nn, hi, wi, ci, co, r, st))
        outs = {}
        for prec in precs:
            shp = ConvShape(nn, hi, wi, ci, co, r, r, st, pad, ho, wo, 1 + _lib.PREC_NAMES[prec])
            shp_f = shp_d = shp_w = shp
            if prec == 'f16x3':                      # f16x3 needs the maxima of its tensor operands (else the launch runs as bf16x6)
                from xas_amd import ops_nn as O
                ax, ad = O.amax_slot_from_value(x.abs().max()), O.amax_slot_from_value(dy.abs().max())
                mk = lambda g, xx: ConvShape(nn, hi, wi, ci, co, r, r, st, pad, ho, wo, 1 + _lib.PREC_NAMES[prec], g.data_ptr(),
                                            xx.data_ptr() if xx is not None else None)
                shp_f, shp_d, shp_w = mk(ax, None), mk(ad, None), mk(ad, ax)
                keep = (ax, ad)
            y = torch.empty_like(dy)
            dx = torch.empty_like(x)
            dw = torch.empty_like(w)
            ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp_w)), device=dev)
            wf, wt = weights_for(shp_f, w, 0), weights_for(shp_d, w, 1)
            runs = {'fwd': lambda: call('xas_conv_fwd', ptr(x), ptr(wf), None, ptr(y), shp_f),
                    'dgrad': lambda: call('xas_conv_dgrad', ptr(dy), ptr(wt), ptr(dx), shp_d),
                    'wgrad': lambda: call('xas_conv_wgrad_oihw', ptr(x), ptr(dy), ptr(dw), ptr(ws), shp_w)}
            line += ' |%s' % prec
            for k, fn in runs.items():
                if which not in ('all', k):
                    continue
                fn()
                if once:
                    continue
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    fn()
                b.record()
                torch.cuda.synchronize()
                ms = a.elapsed_time(b) / reps
                line += ' %s %7.1f us %6.1f TF' % (k, ms * 1e3, flops / ms / 1e9)
                t = tot.setdefault((prec, k), [0.0, 0.0])
                t[0] += flops
                t[1] += ms
            outs[prec] = {'fwd': y, 'dgrad': dx, 'wgrad': dw}
        for pz in precs:
            if pz == 'f32' or 'f32' not in outs:
                continue
            errs = []
            for k in ('fwd', 'dgrad', 'wgrad'):
                if which in ('all', k):
                    a, b = outs[pz][k].double(), outs['f32'][k].double()
                    errs.append('%s %.1e' % (k, float((a - b).norm() / (b.norm() + 1e-30))))
            line += ' | %s vs f32: ' % pz + ' '.join(errs)
        print(line, flush=True)
        del x, dy, w, outs
    for (prec, k), (f, ms) in tot.items():
        print('TOTAL %-7s %-6s %.2f ms  %.1f TF' % (prec, k, ms, f / ms / 1e9))


if __name__ == '__main__':
    main()
