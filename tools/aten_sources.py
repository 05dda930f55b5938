#!/usr/bin/env python3
"""Where do the small ATen kernels of one training step come from?  torch.profiler with Python stacks; every device kernel
that is not one of libxas_hip's is attributed to the innermost frame of this repository.  usage: python tools/aten_sources.py"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from torch.profiler import ProfilerActivity, profile

from xas_amd import engine
from xas_amd.synthetic import model_config, synthetic_batch

cfg = model_config(os.environ.get('WORKLOAD', 'HM36_Multi_SurS1'))
torch.manual_seed(0)
model, disc, od, odisc = engine.prepare_model(cfg)
model.cuda().train(); disc.cuda().train()
step = engine.TrainStep(cfg, model, disc, od, odisc)
x = synthetic_batch(int(os.environ.get('B', 32)), cfg['model_params']['cam_id_list'], torch.device('cuda'), seed=1)
for _ in range(2):
    step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(x)
    torch.cuda.synchronize()
by_site = collections.Counter()
by_site_us = collections.Counter()
for ev in prof.key_averages(group_by_stack_n=12):
    us = getattr(ev, 'self_device_time_total', 0) or 0
    if us <= 0 or not ev.key.startswith('aten::'):
        continue
    st = [f for f in (ev.stack or []) if 'x-as-supervision_amd' in f or '/bench.py' in f]
    site = st[0].split('x-as-supervision_amd/')[-1] if st else '(autograd engine / no repository frame)'
    by_site[(site, ev.key)] += ev.count
    by_site_us[(site, ev.key)] += us
tot = sum(by_site_us.values())
print('ATen ops with device time in one step: %d calls, %.2f ms' % (sum(by_site.values()), tot / 1e3))
for k, us in by_site_us.most_common(70):
    print('%6.0f us %4d  %-28s %s' % (us, by_site[k], k[1], k[0]))
