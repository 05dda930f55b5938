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
    module.load_state_dict(gi.seeded_state_dict(keys, shapes, seed), strict=True)
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
