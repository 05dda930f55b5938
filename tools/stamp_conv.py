#!/usr/bin/env python3
"""Per-phase cycle shares inside the igemm K-loop (diagnostic stamps; shares, not absolute time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from xas_amd._lib import ConvShape, call, ptr, query
dev = torch.device('cuda')
dbg = torch.zeros(8, dtype=torch.int64, device=dev)
for (n, hi, wi, ci, co, r, st, pad) in [(32, 64, 64, 64, 64, 3, 1, 1), (32, 16, 16, 256, 256, 3, 1, 1), (32, 64, 64, 256, 1152, 1, 1, 0), (32, 128, 128, 64, 64, 3, 1, 1)]:
    ho, wo = (hi + 2 * pad - r) // st + 1, (wi + 2 * pad - r) // st + 1
    shp = ConvShape(n, hi, wi, ci, co, r, r, st, pad, ho, wo)
    x = torch.randn(n * hi * wi * ci, device=dev); w = torch.randn(co * r * r * ci, device=dev) * 0.05
    y = torch.empty(n * ho * wo * co, device=dev)
    call('xas_conv_fwd', ptr(x), ptr(w), None, ptr(y), shp)
    dbg.zero_()
    query('xas_set_debug_buffer', dbg.data_ptr())
    call('xas_conv_fwd', ptr(x), ptr(w), None, ptr(y), shp)
    torch.cuda.synchronize()
    query('xas_set_debug_buffer', None)
    d = dbg.cpu().tolist()
    tot = sum(d[:4]); nk = max(1, d[4])
    print('%-34s per K-step per wave: store %5.0f  barrier %5.0f  load-issue %5.0f  mfma %5.0f  (total %5.0f cycles; ideal mfma-only = %d)' % (
        str((n, hi, wi, ci, co, r, st)), d[0] / nk, d[1] / nk, d[2] / nk, d[3] / nk, tot / nk, 4096 if co >= 96 and n*ho*wo*co > 512*128*128 else 1024), flush=True)
