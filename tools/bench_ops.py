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


if __name__ == '__main__':
    main()
