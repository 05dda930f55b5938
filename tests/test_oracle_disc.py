"""The oracle's GCN discriminator (oracle/nets.py: GCNDecouple) against goldens written by the REFERENCE's own
modules/discriminator.py + modules/gcn.py, imported unchanged with only the two torch_geometric primitives restated
(tests/golden/make_golden.py: g_disc).  Pins positional encoding, adjacency, bone vectors, residual order, header."""
import ast

import numpy as np
import pytest
import torch

import inputs as gi
from conftest import golden

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _load(module, g, seed=92):
    keys = g['keys'].tolist()
    shapes = [ast.literal_eval(s) for s in g['shapes'].tolist()]
    assert sorted(module.state_dict().keys()) == sorted(keys)            # same names as the reference (checkpoints load)
    for k, shp in zip(keys, shapes):
        assert list(module.state_dict()[k].shape) == shp, k
    sd = gi.seeded_state_dict(keys, shapes, seed)
    for k, v in module.state_dict().items():                              # batch counters are not seeded: keep the module's zeros
        if k.endswith('num_batches_tracked') and k not in sd:
            sd[k] = v
    module.load_state_dict(sd, strict=True)
    return module


@pytest.mark.parametrize('B', [2, 5])
@pytest.mark.parametrize('mode', ['eval', 'train_p0'])
def test_oracle_decouple_vs_reference_golden(B, mode):
    from oracle.geometry import skeleton_links
    from oracle.nets import GCNDecouple
    g = golden('disc_decouple')
    cfg = gi.model_params('S2')['smpl_disc_params']
    net = _load(GCNDecouple(cfg), g)
    net.parent_ids, net.child_ids = skeleton_links(gi.HM36_PARENTS, gi.LINE_SELECT, False, False)
    net.train(mode != 'eval')
    pre = '%s_B%d_' % (mode, B)
    x = T(g[pre + 'kp']).requires_grad_(True)
    y = net(x)
    assert float((y.detach() - T(g[pre + 'logits'])).abs().max()) < 2e-5 * max(1.0, float(np.abs(g[pre + 'logits']).max()))
    (y * T(g[pre + 'grad_out'])).sum().backward()
    assert rel(x.grad, T(g[pre + 'grad_kp'])) < 2e-5
    p = dict(net.named_parameters())
    for gk, name in (('g_in_w', 'joint_input_layer.weight'), ('g_sage_l', 'joint_gcn.0.gc1.lin_l.weight'),
                     ('g_sage_r', 'joint_gcn.1.gc2.lin_r.weight'), ('g_ln_w', 'joint_gcn.2.ln1.weight'),
                     ('g_ln_b', 'joint_gcn.0.ln2.bias'), ('g_bone_in_b', 'bone_input_layer.bias'),
                     ('g_head2_w', 'header.layer2.weight')):
        assert rel(p[name].grad, T(g[pre + gk])) < 5e-5, name
    assert rel(p['header.layer1.weight'].grad[::16, ::64], T(g[pre + 'g_head1_w_sub'])) < 5e-5


GCN_CASES = [('res', 'res_gcn', False, True), ('res_bn', 'res_gcn', True, True), ('simple_noloop', 'simple_gcn', False, False)]
GCN_GRADS = (('g_in_w', 'input_layer.weight'), ('g_gc0_w', 'gcn.0.gc.lin.weight'), ('g_gc0_b', 'gcn.0.gc.bias'),
             ('g_head_w', 'header.weight'), ('g_res_w', 'gcn.1.gc2.lin.weight'), ('g_last_b', 'gcn.3.gc.bias'),
             ('g_bn_w', 'gcn.2.bn.weight'))


def check_gcn_disc_sequence(net, g, tol_logit=2e-5, tol_grad=5e-5, dev='cpu'):
    """Replays the golden's call sequence (B=2 eval, B=2 train, B=5 eval, B=5 train on ONE module: the batch-norm variant's
    running statistics carry over) and compares logits, input gradient, parameter gradients, running mean."""
    for B in (2, 5):
        for mode in ('eval', 'train_p0'):
            net.train(mode != 'eval')
            net.zero_grad()
            pre = '%s_B%d_' % (mode, B)
            x = T(g[pre + 'kp']).to(dev).requires_grad_(True)
            y = net(x)
            ref = T(g[pre + 'logits'])
            assert float((y.detach().cpu() - ref).abs().max()) < tol_logit * max(1.0, float(ref.abs().max())), pre
            (y * T(g[pre + 'grad_out']).to(dev)).sum().backward()
            assert rel(x.grad.cpu(), T(g[pre + 'grad_kp'])) < tol_grad, pre
            p = dict(net.named_parameters())
            for gk, name in GCN_GRADS:
                if pre + gk in g:
                    assert rel(p[name].grad.cpu(), T(g[pre + gk])) < tol_grad, (pre, name)
            if pre + 'bn_rm' in g:
                assert rel(net.state_dict()['gcn.1.bn.running_mean'].cpu(), T(g[pre + 'bn_rm'])) < 1e-5, pre


@pytest.mark.parametrize('tag,name,use_bn,self_loop', GCN_CASES)
def test_oracle_gcnconv_disc_vs_reference_golden(tag, name, use_bn, self_loop):
    """oracle.nets.GCNConvDisc against the reference's GCNDiscriminator (discriminator.py:80-139) imported unchanged with
    GCNConv restated on edge lists (make_golden.py: g_disc_gcn)."""
    from oracle.geometry import skeleton_links
    from oracle.nets import GCNConvDisc
    g = golden('disc_gcn_' + tag)
    cfg = dict(gi.model_params('S2')['smpl_disc_params'], name=name, use_bn=use_bn, use_self_loop=self_loop)
    net = GCNConvDisc(cfg)
    for m in net.modules():
        if hasattr(m, 'p') and isinstance(getattr(m, 'p'), float):
            m.p = 0.0
    net = _load(net, g, seed=94)
    net.parent_ids, net.child_ids = skeleton_links(gi.HM36_PARENTS, gi.LINE_SELECT, False, False)
    check_gcn_disc_sequence(net, g)
