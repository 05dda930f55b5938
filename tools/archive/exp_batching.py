#!/usr/bin/env python3
"""Experiment: detector forward+backward on 128 images as 4 x B=32, 2 x B=64 or 1 x B=128 passes (what a camera-batched
step would launch).  BN statistics differ between the variants; only the timing is of interest here."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from modules.keypoint_detector_integral_multi import KPDetector3DMulti
from xas_amd import ops_nn
from xas_amd.optim import FusedAdam
torch.manual_seed(0)
det = KPDetector3DMulti('resnet_multi', 18, 64, 3, 15).cuda().train()
opt = FusedAdam(det.parameters(), lr=1e-4, betas=(0.5, 0.999))
opt.grad_arena
total = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.rand(total, 3, 256, 256, device='cuda')
def run(bc, grad=True):
    outs = []
    for i in range(total // bc):
        if grad:
            kps, _ = det(x[i * bc:(i + 1) * bc])
            outs.append(kps.pow(2).mean())
        else:
            with torch.no_grad():
                det(x[i * bc:(i + 1) * bc])
    if grad:
        sum(outs).backward()
        ops_nn.join_side_stream()
        opt.zero_grad()
for grad in (False, True):
    for bc in (32, 64, 128):
        if bc > total: continue
        run(bc, grad); torch.cuda.synchronize()
        ts = []
        for r in range(4):
            t0 = time.perf_counter(); run(bc, grad); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print('grad=%d  %d x B=%-3d : %s  min %.1f ms' % (grad, total // bc, bc, ' '.join('%.1f' % t for t in ts), min(ts)), flush=True)
print('peak GB', torch.cuda.max_memory_allocated() / 2**30)
