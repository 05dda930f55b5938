#!/usr/bin/env python3
"""Do the MFMA convolution kernels write outside their output when they run the discriminator's linear layers
(ops_nn.linear, XAS_LINEAR_MFMA: rows presented as ONE image of H x W pixels)?  Every output sits inside a larger buffer of
sentinels; forward, data gradient, weight gradient (+ bias column sums) for the shapes the step sends."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'x-as-supervision_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch

from xas_amd import ops_nn as F
from xas_amd._lib import call, ptr, query

SENT = 12345.678
PAD = 1 << 20          # floats on either side


def guarded(n, dev='cuda'):
    big = torch.full((n + 2 * PAD,), SENT, device=dev, dtype=torch.float32)
    return big, big[PAD:PAD + n]


def intact(big, n, what):
    lo, hi = big[:PAD], big[PAD + n:]
    bad = int((lo != SENT).sum()) + int((hi != SENT).sum())
    if bad:
        il = (lo != SENT).nonzero().flatten()
        ih = (hi != SENT).nonzero().flatten()
        print('   !!! %s: %d sentinel floats overwritten (below: %d, first at -%d; above: %d, last at +%d)' % (
            what, bad, il.numel(), PAD - int(il.min()) if il.numel() else 0, ih.numel(), int(ih.max()) if ih.numel() else 0))
    return bad == 0


def main():
    ok = True
    for rows, ci, co in ((6912, 128, 128), (9216, 128, 128), (384, 4608, 512), (512, 4608, 512), (6912, 128, 512), (1152, 128, 128),
                         (2304, 128, 128), (768, 4608, 512), (96, 4608, 512), (128, 4608, 512)):
        hw = F._row_map(rows)
        if hw is None:
            print('rows %d: no row map' % rows)
            continue
        g = torch.Generator(device='cuda').manual_seed(rows + ci)
        x = torch.randn(rows, ci, device='cuda', generator=g)
        w = torch.randn(co, ci, device='cuda', generator=g) / ci ** 0.5
        b = torch.randn(co, device='cuda', generator=g)
        dy = torch.randn(rows, co, device='cuda', generator=g)
        shp = F._shape(1, hw[0], hw[1], ci, co, 1, 1, 1, 0, hw[0], hw[1])
        cache = F._PackCache()
        w4 = w.view(co, ci, 1, 1)
        wf, wt = cache.get(w4, 0, shp), cache.get(w4, 1, shp)
        ybig, y = guarded(rows * co)
        call('xas_conv_fwd', ptr(x), ptr(wf), ptr(b), ptr(y), shp)
        torch.cuda.synchronize()
        ok &= intact(ybig, rows * co, 'forward')
        ref = x.double() @ w.double().t() + b.double()
        e_f = float((y.view(rows, co).double() - ref).norm() / ref.norm())
        dxbig, dx = guarded(rows * ci)
        call('xas_conv_dgrad', ptr(dy), ptr(wt), ptr(dx), shp)
        torch.cuda.synchronize()
        ok &= intact(dxbig, rows * ci, 'data gradient')
        ref = dy.double() @ w.double()
        e_d = float((dx.view(rows, ci).double() - ref).norm() / ref.norm())
        dwbig, dw = guarded(co * ci)
        dw.zero_()
        nws = max(1, query('xas_conv_wgrad_workspace_floats', shp))
        wsbig, ws = guarded(nws)
        call('xas_conv_wgrad_acc', ptr(x), ptr(dy), ptr(dw), ptr(ws), shp)
        torch.cuda.synchronize()
        ok &= intact(dwbig, co * ci, 'weight gradient')
        ok &= intact(wsbig, nws, 'weight-gradient workspace (%d floats)' % nws)
        ref = dy.double().t() @ x.double()
        e_w = float((dw.view(co, ci).double() - ref).norm() / ref.norm())
        dbbig, db = guarded(co)
        db.zero_()
        nws = query('xas_bn_workspace_floats', rows, co, 1)
        wsbig, ws = guarded(nws)
        call('xas_col_sum_acc', ptr(dy), rows, co, ptr(db), ptr(ws))
        torch.cuda.synchronize()
        ok &= intact(dbbig, co, 'bias gradient')
        ok &= intact(wsbig, nws, 'column-sum workspace (%d floats)' % nws)
        print('rows %5d (%3d x %3d) %4d -> %4d: fwd %.1e dgrad %.1e wgrad %.1e' % (rows, hw[0], hw[1], ci, co, e_f, e_d, e_w), flush=True)
    print('ALL INTACT' if ok else 'OVERWRITES FOUND')
    pack_split()




def pack_split():
    """The weight-format kernels behind _PackCache.get: xas_pack_weight (transposed) and xas_split_weight, guarded."""
    ok = True
    for co, ci in ((128, 128), (512, 4608), (128, 256), (1152, 256)):
        w = torch.randn(co, ci, 1, 1, device='cuda')
        for transposed in (0, 1):
            pbig, p = guarded(w.numel())
            call('xas_pack_weight', ptr(w), ptr(p), co, ci, 1, 1, transposed)
            torch.cuda.synchronize()
            ok &= intact(pbig, w.numel(), 'pack_weight %dx%d t=%d' % (co, ci, transposed))
            rows = ci if transposed else co
            kk = w.numel() // rows
            for planes in (1, 2, 3):
                nb = query('xas_split_weight_bytes', rows, kk, planes)
                sbig, sp = guarded((nb + 3) // 4)
                call('xas_split_weight', ptr(p), ptr(sp), rows, kk, planes)
                torch.cuda.synchronize()
                ok &= intact(sbig, (nb + 3) // 4, 'split_weight %dx%d t=%d planes=%d' % (co, ci, transposed, planes))
    print('pack / split:', 'INTACT' if ok else 'OVERWRITES FOUND')



if __name__ == '__main__':
    main()
