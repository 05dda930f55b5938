#!/usr/bin/env python3
"""Calibration only: what the vendor fp32 GEMM (torch.matmul -> rocBLAS / hipBLASLt) reaches on GEMMs with the shapes
of the detector's convolutions (M = pixels, N = output channels, K = taps x input channels), warm clocks."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
SHAPES = [(131072, 1152, 256), (131072, 256, 64), (131072, 64, 576), (32768, 128, 1152), (8192, 256, 2304), (8192, 1024, 256),
          (2048, 512, 4608), (2048, 2048, 512), (524288, 64, 576)]
for M, N, K in SHAPES:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(K, N, device='cuda')
    for _ in range(30):
        a @ b
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(20, int(0.15 / (2.0 * M * N * K / 100e12)))
    s.record()
    for _ in range(n):
        a @ b
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print('M=%7d N=%5d K=%5d  %8.1f us  %6.1f TFLOP/s' % (M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
