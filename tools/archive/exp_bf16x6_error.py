#!/usr/bin/env python3
"""Forward-convolution error against float64 for the three precision modes (fp32 MFMA, bf16x6, bf16) on a few layer shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
import torch.nn.functional as TF
from xas_amd import _lib, layers as L


def rel(a, b):
    a, b = a.detach().cpu().double(), b.double()
    return float((a - b).norm() / b.norm())


for (n, cin, h, cout, k, stride, pad) in [(4, 64, 32, 256, 1, 1, 0), (4, 256, 32, 256, 3, 1, 1), (2, 2048, 8, 512, 1, 1, 0),
                                          (4, 512, 8, 512, 3, 1, 1), (2, 256, 16, 256, 4, 2, 1)]:
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, h, h, generator=g) * 2 + 0.3
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    exact = TF.conv2d(x.double(), w.double(), None, stride, pad)
    m = L.Conv2d(cin, cout, k, stride, pad, bias=False).cuda()
    with torch.no_grad():
        m.weight.copy_(w)
    errs = []
    for mode in (0, 2, 1):
        _lib.query('xas_set_precision', mode)
        with torch.no_grad():
            errs.append(rel(m(x.cuda()), exact))
    _lib.query('xas_set_precision', 2)           # library default
    print('K = %5d  %s   fp32 MFMA %.2e   bf16x6 %.2e   bf16 %.2e' % (cin * k * k, (n, cin, h, cout, k, stride), *errs))
