#!/usr/bin/env python3
"""Print the losses of two consecutive optimisation steps, HIP path vs CPU oracle (body of
tests/test_gpu_model.py::test_full_train_step_vs_oracle), for the tree given as argv[1]."""
import os, sys
root = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else '.')
sys.path[:0] = [os.path.join(root, 'tests'), os.path.join(root, 'tests', 'golden'), root, os.path.join(root, 'x-as-supervision_amd')]
import torch
import inputs as gi
import test_gpu_model as t
from modules.discriminator import GCNDiscriminatorDecouple
from modules.model import Counter3DDisc, Counter3DModel
from oracle import step as ostep
from oracle.nets import GCNDecouple
from xas_amd.engine import TrainStep
from xas_amd.optim import FusedAdam
T = t.T
cfg = gi.model_params('S2', cam_ids=(0, 1))
full = {'model_params': cfg, 'train_params': {'lr_kp_detector': 1e-4, 'lr_discriminator': 1e-4}}
reg, phys, oreg, ophys = t._hip_models('S2', (0, 1))
odisc = gi.seeded_fill_(GCNDecouple(cfg['smpl_disc_params']), seed=9)
disc = GCNDiscriminatorDecouple(cfg['smpl_disc_params'])
disc.load_state_dict(odisc.state_dict())
disc.cuda().train()
disc.header.p = 0.0
gen, dis = Counter3DModel(cfg, reg, None, None, phys), Counter3DDisc(cfg, disc, None, None)
odisc.parent_ids, odisc.child_ids = dis.parent_ids, dis.child_ids
opt_det = FusedAdam(list(reg.parameters()) + list(phys.parameters()), lr=1e-4, betas=(0.5, 0.999))
opt_disc = FusedAdam(disc.parameters(), lr=1e-4, betas=(0.5, 0.999))
o_det = torch.optim.Adam(list(oreg.parameters()) + list(ophys.parameters()), lr=1e-4, betas=(0.5, 0.999))
o_disc = torch.optim.Adam(odisc.parameters(), lr=1e-4, betas=(0.5, 0.999))
step = TrainStep(full, gen, dis, opt_det, opt_disc)
xn = gi.synthetic_batch(2, [0, 1], seed=91)
xg = {k: T(v).cuda() for k, v in xn.items()}
xc = {k: T(v) for k, v in xn.items()}
for it in range(int(os.environ.get("NSTEP", "1"))):
    ld, lk, tot, _ = step(xg)
    old, olk = ostep.train_step(cfg, oreg, ophys, odisc, o_det, o_disc, xc)
    print('it=%d  disc %.7f / %.7f' % (it, float(ld), float(old)), ' '.join('%s %.7f/%.7f (%.2e)' % (k, float(lk[k].mean()), float(olk[k]), abs(float(lk[k].mean()) - float(olk[k])) / (abs(float(olk[k])) + 1e-12)) for k in olk), flush=True)
    # parameter distance after the step
    po = torch.cat([p.detach().reshape(-1) for p in odisc.parameters()])
    pg = torch.cat([p.detach().reshape(-1).cpu() for p in disc.parameters()])
    print('      disc params: max|diff| %.3e  frac > 1e-5: %.4f' % (float((po - pg).abs().max()), float(((po - pg).abs() > 1e-5).float().mean())))
print('--- Adam first moments after 3 steps, per discriminator parameter (HIP vs oracle)')
sd = opt_disc.state_dict()['state']
on = dict(odisc.named_parameters())
for i, (name, p) in enumerate(disc.named_parameters()):
    a = sd[i]['exp_avg'].cpu()
    b = o_disc.state[on[name]]['exp_avg']
    r = float((a - b).norm() / (b.norm() + 1e-30))
    if r > 1e-3:
        print('  %-40s rel %.3e  |hip| %.3e |ora| %.3e' % (name, r, float(a.norm()), float(b.norm())))
print('--- detector optimizer')
sd = opt_det.state_dict()['state']
names = [n for n, _ in reg.named_parameters()] + ['phys.' + n for n, _ in phys.named_parameters()]
ops = list(oreg.parameters()) + list(ophys.parameters())
onames = [n for n, _ in oreg.named_parameters()] + ['phys.' + n for n, _ in ophys.named_parameters()]
omap = dict(zip(onames, ops))
bad = 0
for i, name in enumerate(names):
    a = sd[i]['exp_avg'].cpu()
    b = o_det.state[omap[name]]['exp_avg']
    r = float((a - b).norm() / (b.norm() + 1e-30))
    if r > 5e-2:
        bad += 1
        if bad < 40:
            print('  %-50s rel %.3e  |hip| %.3e |ora| %.3e' % (name, r, float(a.norm()), float(b.norm())))
print('params with rel > 5e-2:', bad, 'of', len(names))
