#!/usr/bin/env python3
"""In-process A/B of step-level scheduling knobs (camera streams, wgrad side stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from xas_amd import engine, ops_nn, streams
from xas_amd.synthetic import model_config, synthetic_batch
cfg = model_config('HM36_Multi_SurS1')
torch.manual_seed(0)
model, disc, od, odisc = engine.prepare_model(cfg)
model.cuda().train(); disc.cuda().train()
step = engine.TrainStep(cfg, model, disc, od, odisc)
x = synthetic_batch(32, cfg['model_params']['cam_id_list'], torch.device('cuda'), seed=1)
variants = [('cams=1 side=on', 1, True), ('cams=4 side=on', 4, True), ('cams=2 side=on', 2, True), ('cams=1 side=off', 1, False), ('cams=4 side=off', 4, False)]
for name, n, side in variants:          # warm every variant (allocator pools per stream)
    streams.NUM, ops_nn._side['enabled'] = n, side
    step(x)
torch.cuda.synchronize()
res = {v[0]: [] for v in variants}
for rnd in range(3):
    for name, n, side in variants:
        streams.NUM, ops_nn._side['enabled'] = n, side
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            step(x)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 2 * 1e3)
for k, v in res.items():
    print('%-18s  %s  min %.1f ms' % (k, ' '.join('%.1f' % t for t in v), min(v)), flush=True)
print('peak memory GB', torch.cuda.max_memory_allocated() / 2**30)
