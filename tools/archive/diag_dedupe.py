"""Run the 2-camera S2 step three times (default, default, dedupe) and report where state first differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'x-as-supervision_amd'), os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import torch
import inputs as gi
import test_gpu_model as tm
from modules.discriminator import GCNDiscriminatorDecouple
from modules.model import Counter3DDisc, Counter3DModel
from xas_amd.engine import TrainStep
from xas_amd.optim import FusedAdam

cfg = gi.model_params('S2', cam_ids=(0, 1))
full = {'model_params': cfg, 'train_params': {'lr_kp_detector': 1e-4, 'lr_discriminator': 1e-4}}
xg = {k: torch.from_numpy(v).cuda() for k, v in gi.synthetic_batch(2, [0, 1], seed=93).items()}


def run(dedupe, nsteps=1):
    reg, phys, _, _ = tm._hip_models('S2', (0, 1))
    disc = gi.seeded_fill_(GCNDiscriminatorDecouple(cfg['smpl_disc_params']), seed=9).cuda().train()
    disc.header.p = 0.0
    gen, dis = Counter3DModel(cfg, reg, None, None, phys), Counter3DDisc(cfg, disc, None, None)
    od = FusedAdam(list(reg.parameters()) + list(phys.parameters()), lr=1e-4, betas=(0.5, 0.999))
    odc = FusedAdam(disc.parameters(), lr=1e-4, betas=(0.5, 0.999))
    grads = {}
    orig = od.step
    def spy():
        grads['det'] = od.grad_arena.clone()
        orig()
    od.step = spy
    origd = odc.step
    def spyd():
        grads['disc'] = odc.grad_arena.clone()
        origd()
    odc.step = spyd
    step = TrainStep(full, gen, dis, od, odc, dedupe=dedupe)
    for _ in range(nsteps):
        ld, lk, tot, _ = step(xg)
    torch.cuda.synchronize()
    bufs = {k: v.clone() for k, v in reg.state_dict().items() if 'running' in k}
    return dict(det=od.param_arena.clone(), disc=odc.param_arena.clone(), gdet=grads['det'], gdisc=grads['disc'],
                bufs=bufs, ld=float(ld), tot=float(tot), lk={k: float(v.mean()) for k, v in lk.items()}, names=[n for n, _ in list(reg.named_parameters()) + list(phys.named_parameters())], sizes=[(p.numel() + 3) // 4 * 4 for p in list(reg.parameters()) + list(phys.parameters())])


a, b, c = run(False), run(False), run(True)
for name, u, v in (('default vs default', a, b), ('default vs dedupe', a, c)):
    print('==', name)
    for k in ('det', 'disc', 'gdet', 'gdisc'):
        d = (u[k] - v[k]).abs()
        print(' ', k, 'maxabs diff', float(d.max()), 'n differing', int((d > 0).sum()), '/', d.numel())
    nb = sum(1 for k in u['bufs'] if not torch.equal(u['bufs'][k], v['bufs'][k]))
    print('  buffers differing', nb, '/', len(u['bufs']))
    print('  ld', u['ld'], v['ld'], 'tot', u['tot'], v['tot'])
    for k in u['lk']:
        if u['lk'][k] != v['lk'][k]:
            print('   loss', k, u['lk'][k], v['lk'][k])
    d = (u['gdet'] - v['gdet']).abs()
    if float(d.max()) > 0:
        off = 0
        for n, s in zip(u['names'], u['sizes']):
            m = float(d[off:off + s].max()) if off + s <= d.numel() else -1
            if m > 0:
                print('   grad diff', n, m, 'of', float(u['gdet'][off:off + s].abs().max()))
            off += s
