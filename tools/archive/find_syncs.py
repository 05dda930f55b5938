#!/usr/bin/env python3
"""List the host<->device synchronisation points of one training step (torch.cuda.set_sync_debug_mode('warn')):
every warning is printed with the innermost frames of this repository."""
import os, sys, traceback, warnings, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from xas_amd import engine
from xas_amd.synthetic import model_config, synthetic_batch
cfg = model_config(os.environ.get('WORKLOAD', 'HM36_Multi_SurS1'))
torch.manual_seed(0)
model, disc, od, odisc = engine.prepare_model(cfg)
model.cuda().train(); disc.cuda().train()
step = engine.TrainStep(cfg, model, disc, od, odisc)
x = synthetic_batch(int(os.environ.get('B', 8)), cfg['model_params']['cam_id_list'], torch.device('cuda'), seed=1)
for _ in range(2):
    step(x)
torch.cuda.synchronize()
seen = collections.Counter()


def hook(message, category, filename, lineno, file=None, line=None):
    if 'synchroniz' not in str(message):
        return
    st = [f for f in traceback.extract_stack() if ROOT in f.filename and 'find_syncs' not in f.filename]
    if not st:                                   # no frame of this repository (autograd thread, library code): show the tail
        st = traceback.extract_stack()[-8:-1]
    key = ' <- '.join('%s:%d(%s)' % (os.path.relpath(f.filename, ROOT), f.lineno, f.name) for f in reversed(st[-3:]))
    seen[key] += 1


warnings.showwarning = hook
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
step(x)
torch.cuda.set_sync_debug_mode('default')
torch.cuda.synchronize()
for k, v in seen.most_common():
    print('%3d  %s' % (v, k))
print('total synchronising calls in one step:', sum(seen.values()))
