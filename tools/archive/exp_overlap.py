#!/usr/bin/env python3
"""How much does running two independent detector forward passes on two HIP streams save over running them back to
back?  (Timing experiment only: the passes share batch-norm buffers, so their running statistics race.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from xas_amd import engine
from xas_amd.synthetic import model_config, synthetic_batch
cfg = model_config('HM36_Multi_SurS1')
torch.manual_seed(0)
model, disc, od, odisc = engine.prepare_model(cfg)
model.cuda().train()
reg = model.regressor
G = 4
x = torch.randn(32 * G, 3, 256, 256, device='cuda')
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one(stream):
    with torch.cuda.stream(stream), torch.no_grad():
        reg.forward_groups(x, G)


def run(concurrent, reps=4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one(s1)
        one(s2 if concurrent else s1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(2):
    run(False, 1); run(True, 1)
for _ in range(3):
    print('two passes back to back %.1f ms   on two streams %.1f ms' % (run(False), run(True)), flush=True)
