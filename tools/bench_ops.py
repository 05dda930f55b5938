#!/usr/bin/env python3
"""Standalone timing of the HBM-bound layer kernels at the benchmark shapes (B = 32 x 4 cameras): GB/s of algorithmic
traffic per launch.   python tools/bench_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from xas_amd._lib import call, ptr


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    dev = 'cuda'
    for (n, c, h, w) in [(128, 128, 64, 64), (128, 64, 128, 128)]:
        x = torch.randn(n, h, w, c, device=dev)
        y = torch.empty(n, 2 * h, 2 * w, c, device=dev)
        byts = (x.numel() + y.numel()) * 4
        t = timed(lambda: call('xas_upsample2x_fwd', ptr(x), n, h, w, c, ptr(y)))
        print('upsample2x_fwd  %s  %.3f ms  %.2f TB/s' % ((n, c, h, w), t * 1e3, byts / t / 1e12))
        t = timed(lambda: call('xas_upsample2x_bwd', ptr(y), n, h, w, c, ptr(x)))
        print('upsample2x_bwd  %s  %.3f ms  %.2f TB/s' % ((n, c, h, w), t * 1e3, byts / t / 1e12))


if __name__ == '__main__' and len(sys.argv) == 1:
    main()


def conv_cases():
    """Short-K 1x1 layers of the bottleneck stack: plain forward vs forward + statistics epilogue."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import query
    dev = 'cuda'
    for (n, cin, h, cout, k) in [(256, 64, 64, 256, 1), (256, 128, 32, 512, 1), (256, 256, 64, 64, 1), (256, 64, 64, 64, 3)]:
        x = torch.randn(n, h, h, cin, device=dev)
        w = torch.randn(cout, k, k, cin, device=dev) * 0.05
        y = torch.empty(n, h, h, cout, device=dev)
        shp = F._shape(n, h, h, cin, cout, k, k, 1, k // 2, h, h)
        fl = 2.0 * n * h * h * cin * cout * k * k
        byts = (x.numel() + y.numel()) * 4
        t = timed(lambda: call('xas_conv_fwd', ptr(x), ptr(w), None, ptr(y), shp))
        print('conv_fwd          %s  %.3f ms  %.1f TF  %.2f TB/s' % ((n, cin, h, cout, k), t * 1e3, fl / t / 1e12, byts / t / 1e12))
        G = 8
        mean = torch.empty(G, cout, device=dev); var = torch.empty(G, cout, device=dev)
        ws = torch.empty(query('xas_conv_fwd_bnstats_workspace_floats', shp, G), device=dev)
        t = timed(lambda: call('xas_conv_fwd_bnstats', ptr(x), ptr(w), ptr(y), shp, G, None, ptr(mean), ptr(var), cout, None,
                               ptr(ws), None, None, 0.1))
        print('conv_fwd_bnstats  %s  %.3f ms  %.1f TF  %.2f TB/s' % ((n, cin, h, cout, k), t * 1e3, fl / t / 1e12, byts / t / 1e12))
        for tune in (4, 32, 8388608 | 16777216):
            query('xas_set_tuning', tune)
            t = timed(lambda: call('xas_conv_fwd', ptr(x), ptr(w), None, ptr(y), shp))
            print('   tune %-9d      %.3f ms  %.1f TF' % (tune, t * 1e3, fl / t / 1e12))
        query('xas_set_tuning', 0)


if __name__ == '__main__' and 'conv' in sys.argv[1:]:
    conv_cases()


def thin_cases():
    """Weight gradients of the one-channel 3x3 convs of the physique network (1 -> 32 and 32 -> 1 at 256 x 256)."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import query
    dev = 'cuda'
    n, h, c = 128, 256, 32
    for cin, cout in ((1, c), (c, 1)):
        x = torch.randn(n, h, h, cin, device=dev)
        dy = torch.randn(n, h, h, cout, device=dev)
        shp = F._shape(n, h, h, cin, cout, 3, 3, 1, 1, h, h)
        dw = torch.empty(cout, cin, 3, 3, device=dev)
        ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp)), device=dev)
        t = timed(lambda: call('xas_conv_wgrad_oihw', ptr(x), ptr(dy), ptr(dw), ptr(ws), shp))
        byts = (x.numel() + dy.numel()) * 4
        print('thin wgrad %d -> %d  %.3f ms  %.2f TB/s' % (cin, cout, t * 1e3, byts / t / 1e12))


if __name__ == '__main__' and 'thin' in sys.argv[1:]:
    thin_cases()


def thin_fwd_cases():
    from xas_amd import ops_nn as F
    dev = 'cuda'
    n, h, c = 128, 256, 32
    for cin, cout in ((1, c), (c, 1)):
        x = torch.randn(n, h, h, cin, device=dev)
        y = torch.empty(n, h, h, cout, device=dev)
        w = torch.randn(cout * 9 * cin, device=dev)
        shp = F._shape(n, h, h, cin, cout, 3, 3, 1, 1, h, h)
        byts = (x.numel() + y.numel()) * 4
        t = timed(lambda: call('xas_conv_fwd', ptr(x), ptr(w), None, ptr(y), shp))
        print('thin fwd   %d -> %d  %.3f ms  %.2f TB/s' % (cin, cout, t * 1e3, byts / t / 1e12))
        t = timed(lambda: call('xas_conv_dgrad', ptr(y), ptr(w), ptr(x), shp))
        print('thin dgrad %d -> %d  %.3f ms  %.2f TB/s' % (cin, cout, t * 1e3, byts / t / 1e12))


if __name__ == '__main__' and 'thin' in sys.argv[1:]:
    thin_fwd_cases()


# (per-shape timings in the three precision modes: tools/bench_conv.py)
