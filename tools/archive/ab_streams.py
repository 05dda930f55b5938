#!/usr/bin/env python3
"""A/B of stream layouts for the generator step: weight-gradient side stream on/off x camera streams (set
XAS_CAM_STREAMS in the environment before running)."""
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
res = {}
for side in (True, False):
    ops_nn._side['enabled'] = side
    step(x); torch.cuda.synchronize()
for rnd in range(3):
    for side in (True, False):
        ops_nn._side['enabled'] = side
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            step(x)
        torch.cuda.synchronize()
        res.setdefault(side, []).append((time.perf_counter() - t0) / 2 * 1e3)
for k, v in res.items():
    print('cam streams %d, wgrad side stream %-5s: %s  min %.1f ms' % (streams.NUM, k, ' '.join('%.1f' % t for t in v), min(v)), flush=True)
