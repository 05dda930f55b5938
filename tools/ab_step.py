#!/usr/bin/env python3
"""In-process A/B of step-level scheduling knobs (camera streams, wgrad side stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch
from xas_amd import engine, ops_nn, streams, _lib
from xas_amd.synthetic import model_config, synthetic_batch
cfg = model_config(os.environ.get('WORKLOAD', 'HM36_Multi_SurS1'))
torch.manual_seed(0)
model, disc, od, odisc = engine.prepare_model(cfg)
model.cuda().train(); disc.cuda().train()
step = engine.TrainStep(cfg, model, disc, od, odisc)
x = synthetic_batch(32, cfg['model_params']['cam_id_list'], torch.device('cuda'), seed=1)
# variants: an integer = xas_set_tuning flags; NAME=VALUE = module attribute of xas_amd.ops_nn (e.g. FUSE_DGRAD_BN=0),
# interleaved in ONE process so that box-to-box and clock drift cancel
def parse(a):
    if '=' in a:
        k, v = a.split('=')
        return (a, ('attr', k, int(v)))
    return ('tune=%s' % a, ('tune', int(a)))


def select(sel):
    if sel[0] == 'tune':
        _lib.query('xas_set_tuning', sel[1])
    else:
        _lib.query('xas_set_tuning', 0)
        setattr(ops_nn, sel[1], sel[2] if sel[1].startswith('_') else bool(sel[2]))      # _NAME=int, NAME=0/1


variants = [parse(a) for a in (sys.argv[1:] or ['0', '8192'])]
for name, sel in variants:
    select(sel)
    step(x)
torch.cuda.synchronize()
res = {v[0]: [] for v in variants}
ROUNDS, STEPS = int(os.environ.get('AB_ROUNDS', 5)), int(os.environ.get('AB_STEPS', 3))
for rnd in range(ROUNDS):
    for name, sel in variants:
        select(sel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            step(x)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / STEPS * 1e3)
for k, v in res.items():
    print('%-18s  %s  min %.1f  median %.1f ms' % (k, ' '.join('%.1f' % t for t in v), min(v), sorted(v)[len(v) // 2]), flush=True)
print('peak memory GB', torch.cuda.max_memory_allocated() / 2**30)
