#!/usr/bin/env python3
"""K-loop diagnostics of the igemm kernel (needs the diagnostic library: python tools/build_diag.py, then
XAS_HIP_LIB=x-as-supervision_amd/xas_amd/libxas_hip_diag.so python tools/stamp_conv.py).
Per shape: achieved TFLOP/s with warm clocks for the shipped loop and for ablations of it (no global loads / no LDS
stores+barriers / fragments read once), then the per-phase cycle shares from in-kernel s_memtime stamps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from xas_amd._lib import ConvShape, call, ptr, query
dev = torch.device('cuda')
dbg = torch.zeros(8, dtype=torch.int64, device=dev)
SHAPES = [(32, 64, 64, 256, 1152, 1, 1, 0), (32, 128, 128, 64, 64, 3, 1, 1), (32, 64, 64, 64, 64, 3, 1, 1), (32, 16, 16, 256, 256, 3, 1, 1),
          (32, 32, 32, 128, 512, 1, 1, 0)]
ABL = [(0, 'shipped'), (32, 'pipelined'), (8, 'no loads'), (16, 'no stores/barriers'), (24, 'no loads, no stores'), (24 + 4096, '+ fragments once')]


def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(reps):
        fn()                                   # warm-up: lets the shader clock ramp (tools/micro/mfma_clock)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for (n, hi, wi, ci, co, r, st, pad) in SHAPES:
    ho, wo = (hi + 2 * pad - r) // st + 1, (wi + 2 * pad - r) // st + 1
    shp = ConvShape(n, hi, wi, ci, co, r, r, st, pad, ho, wo)
    x = torch.randn(n * hi * wi * ci, device=dev); w = torch.randn(co * r * r * ci, device=dev) * 0.05
    y = torch.empty(n * ho * wo * co, device=dev)
    fl = 2.0 * n * ho * wo * co * r * r * ci
    fn = lambda: call('xas_conv_fwd', ptr(x), ptr(w), None, ptr(y), shp)
    reps = max(20, int(0.15 / (fl / 90e12)))   # ~150 ms of work per measurement
    line = []
    for tune, tag in ABL:
        query('xas_set_tuning', tune)
        line.append('%s %.1f' % (tag, fl / timed(fn, reps) / 1e9))
    query('xas_set_tuning', 0)
    print('%-32s TFLOP/s: %s' % (str((n, hi, wi, ci, co, r, st)), ' | '.join(line)), flush=True)
    dbg.zero_()
    query('xas_set_debug_buffer', dbg.data_ptr())
    fn()
    torch.cuda.synchronize()
    query('xas_set_debug_buffer', None)
    d = dbg.cpu().tolist()
    tot = sum(d[:4]); nk = max(1, d[4])
    print('    per K-step per wave (cycles): wait+store %5.0f  barrier %5.0f  load-issue %5.0f  mfma %5.0f  total %5.0f' % (
        d[0] / nk, d[1] / nk, d[2] / nk, d[3] / nk, tot / nk), flush=True)
