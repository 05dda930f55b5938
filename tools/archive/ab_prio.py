#!/usr/bin/env python3
"""A/B: main chain on a high-priority stream vs default stream (weight-gradient side stream stays default)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from xas_amd import engine, ops_nn
from xas_amd.synthetic import model_config, synthetic_batch
cfg = model_config('HM36_Multi_SurS1')
torch.manual_seed(0)
model, disc, od, odisc = engine.prepare_model(cfg)
model.cuda().train(); disc.cuda().train()
step = engine.TrainStep(cfg, model, disc, od, odisc)
x = synthetic_batch(32, cfg['model_params']['cam_id_list'], torch.device('cuda'), seed=1)
hi = torch.cuda.Stream(priority=-1)
def run(mode, n=2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if mode == 'hi':
            hi.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(hi):
                step(x)
            torch.cuda.current_stream().wait_stream(hi)
        else:
            step(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for m in ('def', 'hi'):
    run(m, 1)
res = {'def': [], 'hi': []}
for r in range(3):
    for m in ('def', 'hi'):
        res[m].append(run(m))
for k, v in res.items():
    print(k, ' '.join('%.1f' % t for t in v), 'min %.1f' % min(v), flush=True)
