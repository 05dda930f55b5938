#!/usr/bin/env python3
"""Soft-argmax head kernels alone at the BASELINE size (B=32, K=18, D=H=W=64): achieved HBM GB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from xas_amd import ops_head
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = 20
lg = torch.randn(B, 64, 64, 1152, device='cuda').permute(0, 3, 1, 2).requires_grad_(True)   # NHWC storage
g = torch.randn(B, 3, 18, 3, device='cuda')
kps, _, _ = ops_head.softargmax_multi(lg, 18, 3, 15)
kps.backward(g)
torch.cuda.synchronize()
a, b, c = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(reps):
    lg.grad = None
    a.record()
    kps, _, _ = ops_head.softargmax_multi(lg, 18, 3, 15)
    b.record()
    kps.backward(g)
    c.record()
    torch.cuda.synchronize()
    tf += a.elapsed_time(b); tb += b.elapsed_time(c)
nbytes = B * 1152 * 64 * 64 * 4
print('head fwd  %.1f us  %.0f GB/s (reads %d MB once)' % (tf / reps * 1e3, nbytes / (tf / reps * 1e-3) / 1e9, nbytes >> 20))
print('head bwd  %.1f us  %.0f GB/s (read + write)' % (tb / reps * 1e3, 2 * nbytes / (tb / reps * 1e-3) / 1e9))
