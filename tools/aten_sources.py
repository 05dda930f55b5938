#!/usr/bin/env python3
"""Where do the small ATen kernels of one training step come from?  A TorchDispatchMode counts every ATen call of one step
with (a) the innermost frame of this repository for forward calls and (b) the autograd node being executed for backward
calls.  usage: python tools/aten_sources.py   (GPU box)"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from xas_amd import engine
from xas_amd.synthetic import model_config, synthetic_batch

SKIP = ('aten.view', 'aten.detach', 'aten.slice', 'aten.select', 'aten.as_strided', 'aten._unsafe_view', 'aten.t.', 'aten.alias',
        'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.permute', 'aten.transpose', 'aten.reshape', 'aten.empty',
        'aten.sym_', 'aten.is_', 'aten.size', 'aten.stride', 'aten.unbind', 'aten.split', 'aten._local_scalar', 'aten.lift_fresh',
        'aten.new_empty', 'aten.empty_like', 'aten.unfold', 'aten.narrow', 'aten.chunk', 'aten.result_type', 'aten.item',
        'aten.contiguous', 'prim.', 'aten.set_', 'aten.resize_', 'aten.record_stream')


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            node = torch._C._current_autograd_node()
            if node is not None:
                site = 'bwd:' + type(node).__name__
            else:
                st = [f for f in traceback.extract_stack() if 'x-as-supervision_amd' in f.filename]
                site = ('%s:%d' % (os.path.relpath(st[-1].filename, ROOT).replace('x-as-supervision_amd/', ''), st[-1].lineno)) if st else '?'
            self.by[(name, site)] += 1
        return func(*args, **(kwargs or {}))


cfg = model_config(os.environ.get('WORKLOAD', 'HM36_Multi_SurS1'))
torch.manual_seed(0)
model, disc, od, odisc = engine.prepare_model(cfg)
model.cuda().train(); disc.cuda().train()
step = engine.TrainStep(cfg, model, disc, od, odisc)
x = synthetic_batch(int(os.environ.get('B', 32)), cfg['model_params']['cam_id_list'], torch.device('cuda'), seed=1)
for _ in range(2):
    step(x)
torch.cuda.synchronize()
with Count() as c:
    step(x)
torch.cuda.synchronize()
print('ATen calls that may launch device work in one step: %d' % sum(c.by.values()))
by_site = collections.Counter()
for (name, site), n in c.by.items():
    by_site[site] += n
print('--- by site')
for site, n in by_site.most_common(45):
    ops = collections.Counter({k[0]: v for k, v in c.by.items() if k[1] == site})
    print('%5d  %-55s %s' % (n, site, ', '.join('%s x%d' % (k.replace('aten.', ''), v) for k, v in ops.most_common(5))))
