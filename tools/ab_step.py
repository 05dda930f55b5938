#!/usr/bin/env python3
"""In-process A/B of step-level scheduling knobs (camera streams, wgrad side stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from xas_amd import engine, ops_nn, streams, _lib
from xas_amd.synthetic import model_config, synthetic_batch
cfg = model_config('HM36_Multi_SurS1')
torch.manual_seed(0)
model, disc, od, odisc = engine.prepare_model(cfg)
model.cuda().train(); disc.cuda().train()
step = engine.TrainStep(cfg, model, disc, od, odisc)
x = synthetic_batch(32, cfg['model_params']['cam_id_list'], torch.device('cuda'), seed=1)
variants = [('tune=0', 0), ('tune=8192', 8192)] if len(sys.argv) < 2 else [('tune=%s' % t, int(t)) for t in sys.argv[1:]]
for name, t in variants:
    _lib.query('xas_set_tuning', t)
    step(x)
torch.cuda.synchronize()
res = {v[0]: [] for v in variants}
for rnd in range(3):
    for name, t in variants:
        _lib.query('xas_set_tuning', t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            step(x)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 2 * 1e3)
for k, v in res.items():
    print('%-18s  %s  min %.1f ms' % (k, ' '.join('%.1f' % t for t in v), min(v)), flush=True)
print('peak memory GB', torch.cuda.max_memory_allocated() / 2**30)
