import sys, os
sys.path[:0] = ['/root/repo', '/root/repo/x-as-supervision_amd', '/root/repo/tests', '/root/repo/tests/golden']
import numpy as np, torch
import inputs as gi
from conftest import golden
from test_gpu_nn import _hip_regressor, rel, T
g = golden('detector')
hip, ora = _hip_regressor(True)
hip.train(); ora.train()
x = T(gi.synthetic_batch(2, [0], seed=63)['cam_0_img'])
kps, dmap = hip(x.cuda())
(kps * T(g['grad_out']).cuda()).sum().backward()
ko, do = ora(x)
(ko * T(g['grad_out'])).sum().backward()
po = dict(ora.named_parameters())
worst = []
for n, p in hip.named_parameters():
    r = rel(p.grad, po[n].grad)
    worst.append((r, n))
worst.sort(reverse=True)
for r, n in worst[:25]: print('%.3e %s' % (r, n))
print('median', np.median([w[0] for w in worst]))
print('kps maxabs', float((kps.cpu()-ko).abs().max()))
for k in ['g_conv1','g_fin_b','g_bn1_w']:
    pass
print('conv1 vs golden: hip %.3e oracle %.3e' % (rel(dict(hip.named_parameters())['net.backbone.conv1.weight'].grad, T(g['g_conv1'])), rel(po['net.backbone.conv1.weight'].grad, T(g['g_conv1']))))
# fp64 ground truth: are both fp32 implementations equally far from it?
import copy
o64 = copy.deepcopy(ora).double()
o64.zero_grad()
k64, _ = o64(x.double())
(k64 * T(g['grad_out']).double()).sum().backward()
p64 = dict(o64.named_parameters())
eh = [rel(p.grad, p64[n].grad) for n, p in hip.named_parameters()]
eo = [rel(po[n].grad, p64[n].grad) for n, p in hip.named_parameters()]
print('vs fp64: hip median %.3e max %.3e | oracle-fp32 median %.3e max %.3e' % (np.median(eh), max(eh), np.median(eo), max(eo)))
print('kps vs fp64: hip %.3e oracle32 %.3e' % (float((kps.detach().cpu().double()-k64.detach()).abs().max()), float((ko.detach().double()-k64.detach()).abs().max())))
