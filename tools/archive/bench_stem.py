"""Micro-benchmark of the 7x7 stride-2 stem forward at 256 images: the f16x3 kernel against the exact-fp32 kernel (tune
bit 25) and a float64 convolution.  usage: python tools/bench_stem.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
from xas_amd import _lib
from xas_amd._lib import ConvShape, call, ptr, query
import torch.nn.functional as TF
torch.manual_seed(0)
n, h = 256, 256
x = (torch.randn(n, h, h, 3, device='cuda') * 1.2)
w = torch.randn(64, 7, 7, 3, device='cuda') / 147 ** 0.5
outs = {}
for prec, tune in (('f16x3', 0), ('f16x3', 1 << 25)):
    query('xas_set_tuning', tune)
    shp = ConvShape(n, h, h, 3, 64, 7, 7, 2, 3, 128, 128, 1 + _lib.PREC_NAMES[prec])
    y = torch.empty(n, 128, 128, 64, device='cuda')
    f = lambda: call('xas_conv_fwd', ptr(x), ptr(w), None, ptr(y), shp)
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(prec, tune, '%.1f us  %.1f TF' % (ms * 1e3, 2.0 * n * 128 * 128 * 64 * 147 / ms / 1e9))
    outs[tune] = y.clone()
query('xas_set_tuning', 0)
xs = x[:8].permute(0, 3, 1, 2).double().cpu(); ws = w.permute(0, 3, 1, 2).double().cpu()
ref = TF.conv2d(xs, ws, None, 2, 3).permute(0, 2, 3, 1)
for t, y in outs.items():
    e = (y[:8].double().cpu() - ref).norm() / ref.norm()
    print('tune', t, 'rel err vs float64: %.2e' % float(e))
