#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one step (no synchronisation) vs the GPU to finish it?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from xas_amd import engine
from xas_amd.synthetic import model_config, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = model_config('HM36_Multi_SurS1')
torch.manual_seed(0)
model, disc, od, odisc = engine.prepare_model(cfg)
model.cuda().train(); disc.cuda().train()
step = engine.TrainStep(cfg, model, disc, od, odisc)
x = synthetic_batch(B, cfg['model_params']['cam_id_list'], torch.device('cuda'), seed=1)
for _ in range(2):
    step(x)
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    step(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('B=%d enqueue %.1f ms, GPU finished after %.1f ms' % (B, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
