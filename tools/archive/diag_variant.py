"""Train-step variant check in detail: from ONE state, the same step in f16x3, bf16x6, exact fp32 (twice: run-to-run noise of the
exact path itself) - loss terms, gradient-arena differences, launches of xas_abs_max per step.   python tools/diag_variant.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'x-as-supervision_amd')):
    sys.path.insert(0, p)
import torch                                             # noqa: E402
import bench                                             # noqa: E402
from xas_amd import _lib as xl, engine, ops_nn           # noqa: E402
from xas_amd.synthetic import model_config, synthetic_batch   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = sys.argv[2] if len(sys.argv) > 2 else 'HM36_Multi_SurS1'
cfg = model_config(wl)
torch.manual_seed(1234)
model, disc, od, odisc = engine.prepare_model(cfg)
model.cuda().train(); disc.cuda().train()
step = engine.TrainStep(cfg, model, disc, od, odisc)
x = synthetic_batch(B, cfg['model_params']['cam_id_list'], torch.device('cuda'), seed=100)
for _ in range(2):
    step(x)
torch.cuda.synchronize()
n0 = ops_nn.amax_stats['abs_max']
step(x)
print('xas_abs_max launches in one step:', ops_nn.amax_stats['abs_max'] - n0)
for mode in ('f16x3', 'bf16x6', 'f32'):
    r = bench.train_step_variant_check(step, x, model, disc, od, odisc, xl, mode)
    print(mode, 'vs f32: max loss rel diff %.3e' % r['max_loss_rel_diff'], 'grad arena rel diff', r['grad_arena_rel_diff'])
    print('   ', {k: '%.2e' % v for k, v in r['loss_rel_diff'].items()})
