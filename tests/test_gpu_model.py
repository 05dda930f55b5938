"""GPU parity of the physique net, GCN discriminator, SMPL layer, fused Adam, the model wiring and the
full optimisation step (HIP path through the C ABI) against goldens / the CPU oracle."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

import inputs as gi
from conftest import golden

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def maxabs(a, b):
    return float((a.detach().cpu() - b.detach().cpu()).abs().max())


def test_physique_net_vs_golden():
    from modules.physique_network import PhysiqueMaskGenerator
    from oracle.nets import PhysiqueNet
    g = golden('physique')
    ora = gi.seeded_fill_(PhysiqueNet([32, 64, 128]), seed=51)
    net = PhysiqueMaskGenerator([32, 64, 128])
    assert list(net.state_dict().keys()) == g['keys'].tolist()
    net.load_state_dict(ora.state_dict(), strict=True)
    net.cuda().train()
    x = T(g['x']).cuda().requires_grad_(True)
    y = net(x)
    assert maxabs(y, T(g['y'])) < 5e-6
    (y * T(g['grad_out']).cuda()).sum().backward()
    assert rel(x.grad, T(g['grad_x'])) < 2e-4
    assert rel(net.encoder[0][0].weight.grad, T(g['grad_enc0_w'])) < 2e-4
    assert rel(net.decoder[4].bias.grad, T(g['grad_dec4_b'])) < 2e-4
    assert abs(float(net.decoder[1][1].weight.grad.norm()) / float(g['grad_dec1_w_norm']) - 1) < 2e-4
    sd = net.state_dict()
    assert maxabs(sd['encoder.0.1.running_mean'], T(g['run_mean_enc0'])) < 1e-6
    assert maxabs(sd['encoder.0.1.running_var'], T(g['run_var_enc0'])) < 1e-6


def test_physique_net_all_parameter_gradients():
    """EVERY parameter gradient of the physique net against the reference's (golden physique_allgrads: norms + strided samples
    of its fp32 run, and its own fp32-vs-fp64 distance per tensor): per-tensor tolerance max(2e-4, 4 x that distance).  The
    biases of the nine convolutions that feed a norm have no gradient in exact arithmetic (fp64: 1e-14): noise only."""
    from conftest import check_all_grads
    from modules.physique_network import PhysiqueMaskGenerator
    from oracle.nets import PhysiqueNet
    g = golden('physique_allgrads')
    net = PhysiqueMaskGenerator([32, 64, 128])
    net.load_state_dict(gi.seeded_fill_(PhysiqueNet([32, 64, 128]), seed=151).state_dict(), strict=True)
    net.cuda().train()
    y = net(T(g['x']).cuda())
    assert maxabs(y, T(g['y'])) < 5e-6
    (y * T(g['grad_out']).cuda()).sum().backward()
    names = g['names'].tolist()
    noise = {n for n, d in zip(names, g['dev']) if float(d) > 1.0}
    assert len(noise) == 9 and all(n.endswith('.bias') for n in noise)
    worst = check_all_grads([(n, p.grad) for n, p in net.named_parameters()], g, 2e-4, 4.0, 'physique', noise_only=noise)
    print('physique net: worst error / tolerance over %d tensors: %.2f' % (len(names) - len(noise), worst))


def test_decouple_discriminator_all_parameter_gradients():
    """EVERY parameter gradient of GCNDiscriminatorDecouple against the reference's own module (golden
    disc_decouple_allgrads, train mode, dropout 0): per-tensor tolerance max(1e-4, 4 x the reference's fp32-vs-fp64 distance)."""
    import ast
    from conftest import check_all_grads
    from modules.discriminator import GCNDiscriminatorDecouple
    from modules.model import cal_links
    g = golden('disc_decouple_allgrads')
    hip = GCNDiscriminatorDecouple(gi.model_params('S2')['smpl_disc_params'])
    keys = g['keys'].tolist()
    shapes = [ast.literal_eval(s) for s in g['shapes'].tolist()]
    hip.load_state_dict(gi.seeded_state_dict(keys, shapes, 192), strict=True)
    hip.parent_ids, hip.child_ids = cal_links(gi.HM36_PARENTS, gi.LINE_SELECT, use_root=False, extension=False)
    hip.cuda().train()
    hip.header.p = 0.0
    y = hip(T(g['kp']).cuda())
    ref = T(g['y'])
    assert maxabs(y, ref) < 2e-5 * max(1.0, float(ref.abs().max()))
    (y * T(g['grad_out']).cuda().reshape(y.shape)).sum().backward()
    named = dict(hip.named_parameters())
    worst = check_all_grads([(n, named[n].grad) for n in g['names'].tolist()], g, 1e-4, 4.0, 'decouple discriminator')
    print('decouple discriminator: worst error / tolerance over %d tensors: %.2f' % (len(named), worst))


def _discs(seed=7):
    from modules.discriminator import GCNDiscriminatorDecouple
    from modules.model import cal_links
    from oracle.nets import GCNDecouple
    cfg = gi.model_params('S2')['smpl_disc_params']
    ora = gi.seeded_fill_(GCNDecouple(cfg), seed=seed, gain=1.0)
    hip = GCNDiscriminatorDecouple(cfg)
    assert sorted(hip.state_dict().keys()) == sorted(ora.state_dict().keys())
    hip.load_state_dict(ora.state_dict(), strict=True)
    p, c = cal_links(gi.HM36_PARENTS, gi.LINE_SELECT, use_root=False, extension=False)
    ora.parent_ids, ora.child_ids = p, c
    hip.parent_ids, hip.child_ids = p, c
    return hip.cuda(), ora


@pytest.mark.parametrize('B', [2, 32])
def test_gcn_discriminator_vs_oracle(B):
    hip, ora = _discs()
    g = torch.Generator().manual_seed(B)
    kp = torch.randn(B, 18, 3, generator=g) * 0.5
    keep = (torch.rand(B, 512, generator=g) >= 0.2).float()
    kc = kp.clone().requires_grad_(True)
    yo = ora(kc, drop_mask=keep)
    gy = torch.randn(B, 1, generator=g)
    (yo * gy).sum().backward()
    hip.train()
    hip.header.drop_mask = keep.cuda()
    kg = kp.cuda().requires_grad_(True)
    yg = hip(kg)
    assert yg.shape == (B, 1)
    (yg * gy.cuda()).sum().backward()
    assert maxabs(yg, yo) < 2e-5 * max(1.0, float(yo.abs().max()))
    assert rel(kg.grad, kc.grad) < 1e-4
    po = dict(ora.named_parameters())
    for n, p in hip.named_parameters():
        assert rel(p.grad, po[n].grad) < 2e-4, n
    hip.eval()
    assert maxabs(hip(kp.cuda()), ora(kp)) < 2e-5 * max(1.0, float(yo.abs().max()))


@pytest.mark.parametrize('tag', ['decouple', 'sage'])
def test_gcn_discriminators_vs_reference_golden(tag):
    """HIP GCNDiscriminatorDecouple / GCNSAGEDiscriminator against goldens written by the REFERENCE's own
    modules/discriminator.py + modules/gcn.py (imported unchanged, the two torch_geometric primitives restated:
    tests/golden/make_golden.py g_disc): logits, input gradient, parameter gradients, eval and train (dropout 0)."""
    import ast
    from modules.discriminator import GCNDiscriminatorDecouple, GCNSAGEDiscriminator
    from modules.model import cal_links
    g = golden('disc_' + tag)
    cfg = gi.model_params('S2')['smpl_disc_params']
    hip = (GCNDiscriminatorDecouple if tag == 'decouple' else GCNSAGEDiscriminator)(cfg)
    keys = g['keys'].tolist()
    shapes = [ast.literal_eval(s) for s in g['shapes'].tolist()]
    assert sorted(hip.state_dict().keys()) == sorted(keys)               # reference checkpoints load unchanged
    hip.load_state_dict(gi.seeded_state_dict(keys, shapes, 92), strict=True)
    hip.parent_ids, hip.child_ids = cal_links(gi.HM36_PARENTS, gi.LINE_SELECT, use_root=False, extension=False)
    hip.cuda()
    if tag == 'decouple':
        hip.header.p = 0.0
    gcn = 'joint_gcn' if tag == 'decouple' else 'gcn'
    first = 'joint_input_layer.weight' if tag == 'decouple' else 'input_layer.weight'
    probes = [('g_in_w', first), ('g_sage_l', gcn + '.0.gc1.lin_l.weight'), ('g_sage_r', gcn + '.1.gc2.lin_r.weight'),
              ('g_ln_w', gcn + '.2.ln1.weight'), ('g_ln_b', gcn + '.0.ln2.bias')]
    probes += [('g_bone_in_b', 'bone_input_layer.bias'), ('g_head2_w', 'header.layer2.weight')] if tag == 'decouple' \
        else [('g_head_w', 'header.weight')]
    for B in (2, 5):
        for mode in ('eval', 'train_p0'):
            hip.train(mode != 'eval')
            hip.zero_grad()
            pre = '%s_B%d_' % (mode, B)
            x = T(g[pre + 'kp']).cuda().requires_grad_(True)
            y = hip(x)
            ref = T(g[pre + 'logits'])
            assert maxabs(y, ref) < 2e-5 * max(1.0, float(ref.abs().max())), (tag, pre)
            (y * T(g[pre + 'grad_out']).cuda()).sum().backward()
            assert rel(x.grad, T(g[pre + 'grad_kp'])) < 1e-4, (tag, pre)
            p = dict(hip.named_parameters())
            for gk, name in probes:
                assert rel(p[name].grad, T(g[pre + gk])) < 2e-4, (tag, pre, name)
            if tag == 'decouple':
                assert rel(p['header.layer1.weight'].grad[::16, ::64], T(g[pre + 'g_head1_w_sub'])) < 2e-4


def test_discriminator_groups_equal_separate_calls():
    """forward_groups == one call per input (graph-LayerNorm statistics stay per input), values and grads."""
    hip, _ = _discs(seed=11)
    hip.train()
    hip.header.p = 0.0
    g = torch.Generator().manual_seed(5)
    xs = [(torch.randn(32, 18, 3, generator=g) * (0.2 + i)).cuda().requires_grad_(True) for i in range(5)]
    sep = [hip(x) for x in xs]
    torch.stack(sep).pow(2).sum().backward()
    gsep = [x.grad.clone() for x in xs]
    psep = {n: p.grad.clone() for n, p in hip.named_parameters()}
    hip.zero_grad()
    for x in xs:
        x.grad = None
    grp = hip.forward_groups(xs)
    torch.stack(grp).pow(2).sum().backward()
    for a, b in zip(sep, grp):
        assert maxabs(a, b) < 1e-5 * max(1.0, float(a.abs().max()))
    for x, gs in zip(xs, gsep):
        assert rel(x.grad, gs) < 1e-4
    for n, p in hip.named_parameters():
        assert rel(p.grad, psep[n]) < 1e-4, n


def test_fused_pose_losses_vs_oracle_and_golden():
    """Fused min-over-hypotheses losses and LSGAN terms vs the oracle's loss functions (values and gradients),
    and vs the reference goldens for the single-hypothesis forms."""
    from modules.base_losses import loss_func as LF
    from oracle import losses as L
    g = golden('losses')
    gen = torch.Generator().manual_seed(3)
    # goldens: Hy = 1 forms
    kp3, kp2 = T(g['kp3']), T(g['kp2'])
    v = LF.compute_symmetry_min(kp3.unsqueeze(1).cuda(), 1.0, 0.0)
    assert abs(float(v) - float(g['bone_sym'])) < 1e-6 + 1e-5 * abs(float(g['bone_sym']))
    v = LF.compute_symmetry_min(kp3.unsqueeze(1).cuda(), 0.0, 1.0)
    assert abs(float(v) - float(g['kp_sym3'])) < 1e-6 + 1e-5 * abs(float(g['kp_sym3']))
    v = LF.compute_supervision_min(kp3.unsqueeze(1).cuda(), kp3.flip(0).cuda())
    assert abs(float(v) - float(g['sup'])) < 1e-1 + 1e-5 * abs(float(g['sup']))
    assert abs(float(LF.compute_disc_loss(T(g['lg3']).cuda(), None)) - float(g['disc_gen3'])) < 1e-6
    assert abs(float(LF.compute_disc_loss(T(g['lg2']).cuda(), None)) - float(g['disc_gen2'])) < 1e-6
    assert abs(float(LF.compute_disc_loss(T(g['lg3']).cuda(), T(g['gt2']).cuda())) - float(g['disc_d'])) < 1e-6
    # multi-hypothesis min + gradients vs the oracle composition
    B, Hy, K = 32, 3, 18
    world = torch.randn(B, Hy, K, 3, generator=gen) * 400
    kps = torch.randn(B, Hy, K, 3, generator=gen) * 0.5
    gt = torch.randn(B, K, 3, generator=gen) * 0.5
    for wb, wk, w2 in ((0.1, 0.1, 0.0), (0.3, 0.05, 0.5)):
        wc, kc = world.clone().requires_grad_(True), kps.clone().requires_grad_(True)
        ref = torch.stack([L.bone_sym(wc[:, h]) * wb + L.kp_sym(wc[:, h]) * wk + L.kp_sym(kc[:, h, :, :2], False) * 1e2 * w2
                           for h in range(Hy)]).min()
        ref.backward()
        wg, kg = world.cuda().requires_grad_(True), kps.cuda().requires_grad_(True)
        out = LF.compute_symmetry_min(wg, wb, wk, kg, w2)
        out.backward()
        assert abs(float(out) - float(ref)) < 1e-6 + 1e-5 * abs(float(ref))
        assert rel(wg.grad, wc.grad) < 1e-4
        if w2:
            assert rel(kg.grad, kc.grad) < 1e-4
    pc = kps.clone().requires_grad_(True)
    ref = torch.stack([L.supervision(pc[:, h], gt) for h in range(Hy)]).min()
    ref.backward()
    pg = kps.cuda().requires_grad_(True)
    out = LF.compute_supervision_min(pg, gt.cuda())
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-6 and rel(pg.grad, pc.grad) < 1e-5
    lg = torch.randn(B, Hy, 1, generator=gen)
    rl = torch.randn(B, 1, generator=gen)
    lc, rc = lg.clone().requires_grad_(True), rl.clone().requires_grad_(True)
    ref = L.disc_loss(lc, rc)
    ref.backward()
    lgp, rgp = lg.cuda().requires_grad_(True), rl.cuda().requires_grad_(True)
    out = LF.compute_disc_loss(lgp, rgp)
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-6 and rel(lgp.grad, lc.grad) < 1e-5 and rel(rgp.grad, rc.grad) < 1e-5


def test_smpl_layer_vs_golden():
    from modules.smplpytorch.pytorch.smpl_layer import SMPL_Layer
    from modules.util import smpl_to_h36m
    g = golden('smpl')
    buf = gi.smpl_buffers(seed=71)
    lay = SMPL_Layer.from_arrays(buf, center_idx=0).cuda()
    verts, jtr = lay(T(g['pose']).cuda(), T(g['betas']).cuda())
    assert verts.shape == (3, 6890, 3) and jtr.shape == (3, 24, 3)
    assert maxabs(verts[:, ::10], T(g['verts_sub'])) < 3e-5
    assert maxabs(jtr, T(g['joints'])) < 3e-5
    assert maxabs(smpl_to_h36m(verts, T(buf['h36m_regressor']).cuda()), T(g['h36m'])) < 3e-5
    assert 'th_posedirs' in lay.state_dict() and 'th_weights' in lay.state_dict()


@pytest.mark.parametrize('center', [0, 5, None])
def test_smpl_layer_backward_vs_oracle_autograd(center):
    """d(pose), d(betas) of the SMPL layer (xas_smpl_lbs_bwd) against float64 autograd through the oracle's restatement
    of smpl_layer.py:63-156 (itself pinned by the reference-import golden above): random output gradients on verts and
    joints, a near-zero rotation among the joints (the 1e-8 of rodrigues_layer.py:41), deterministic run to run."""
    from modules.smplpytorch.pytorch.smpl_layer import SMPL_Layer
    from oracle import smpl as osmpl
    buf = gi.smpl_buffers(seed=71)
    gen = torch.Generator().manual_seed(5)
    B = 3
    pose = 0.6 * torch.randn(B, 72, generator=gen)
    pose[0, 9:12] = 0.0                                   # identity rotation: angle = |1e-8|
    pose[1, 30:33] = torch.tensor([1e-4, -2e-4, 5e-5])
    betas = torch.randn(B, 10, generator=gen)
    gv = torch.randn(B, 6890, 3, generator=gen) / 6890 ** 0.5
    gj = torch.randn(B, 24, 3, generator=gen)
    pd_, bd = pose.double().requires_grad_(True), betas.double().requires_grad_(True)
    D = lambda k: T(buf[k]).double()
    v_ref, j_ref = osmpl.smpl_lbs(pd_, bd, D('v_template'), D('shapedirs'), D('posedirs'), D('J_regressor'), D('weights'),
                                  center_idx=center)
    ((v_ref * gv.double()).sum() + (j_ref * gj.double()).sum()).backward()

    lay = SMPL_Layer.from_arrays(buf, center_idx=center).cuda()
    grads = []
    for _ in range(2):
        pg, bg = pose.cuda().requires_grad_(True), betas.cuda().requires_grad_(True)
        verts, jtr = lay(pg, bg)
        ((verts * gv.cuda()).sum() + (jtr * gj.cuda()).sum()).backward()
        grads.append((pg.grad.clone(), bg.grad.clone()))
    assert maxabs(verts, v_ref.float()) < 3e-5 and maxabs(jtr, j_ref.float()) < 3e-5
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])
    mask = torch.ones(B, 72, dtype=torch.bool)
    mask[0, 9:12] = False                                 # d/dtheta at theta = 0 divides by the 1e-8 guard: compare the rest tightly
    ref_p, got_p = pd_.grad.float(), grads[0][0].cpu()
    assert float((got_p - ref_p)[mask].abs().max()) < 2e-4 * float(ref_p[mask].abs().max())
    assert rel(grads[0][1], bd.grad) < 2e-4
    # only the vertex gradient / only the joint gradient
    pg, bg = pose.cuda().requires_grad_(True), betas.cuda().requires_grad_(True)
    verts, jtr = lay(pg, bg)
    (jtr * gj.cuda()).sum().backward()
    p2, b2 = pose.double().requires_grad_(True), betas.double().requires_grad_(True)
    _, j2 = osmpl.smpl_lbs(p2, b2, D('v_template'), D('shapedirs'), D('posedirs'), D('J_regressor'), D('weights'),
                           center_idx=center)
    (j2 * gj.double()).sum().backward()
    assert float((pg.grad.cpu() - p2.grad.float())[mask].abs().max()) < 2e-4 * float(p2.grad[mask].abs().max())
    assert rel(bg.grad, b2.grad) < 2e-4


def test_fused_adam_vs_torch():
    from xas_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 3, 7, 7), (64,), (17, 5), (1,), (128, 64, 3, 3)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    a = [nn.Parameter(p.clone().cuda()) for p in ps]
    b = [nn.Parameter(p.clone()) for p in ps]
    oa = FusedAdam(a, lr=2e-4, betas=(0.5, 0.999))
    ob = torch.optim.Adam(b, lr=2e-4, betas=(0.5, 0.999))
    for it in range(3):
        for pa, pb in zip(a, b):
            gr = torch.randn(pb.shape, generator=g) * (0.1 + it)
            pb.grad = gr.clone()
            if pa.grad is None:
                oa.zero_grad()
            pa.grad.copy_(gr.cuda())
        oa.step()
        ob.step()
        oa.zero_grad()
        ob.zero_grad()
    for pa, pb in zip(a, b):
        assert maxabs(pa, pb) < 2e-7
        assert float(pa.grad.abs().max()) == 0.0
    sd = oa.state_dict()
    assert set(sd['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
    assert maxabs(sd['state'][4]['exp_avg'], ob.state_dict()['state'][4]['exp_avg']) < 1e-6


class LinearDisc(nn.Module):
    name = 'LinearStandIn'

    def __init__(self):
        super().__init__()
        self.fc = nn.Linear(54, 1)

    def forward(self, kp):
        return self.fc(kp.reshape(kp.shape[0], -1))


def _hip_models(stage, cam_ids):
    from modules.keypoint_detector_integral_multi import KPDetector3DMulti
    from modules.physique_network import PhysiqueMaskGenerator
    from oracle import step as ostep
    from oracle.nets import PhysiqueNet
    ora_reg = gi.seeded_fill_(ostep.Regressor('resnet_multi', 18, 64, 3, 15), seed=61)
    with torch.no_grad():
        ora_reg.net.head.features[9].bias.copy_(T(gi.planted_depth_bias(18, 64, seed=62)))
    ora_phys = gi.seeded_fill_(PhysiqueNet([32, 64, 128]), seed=81)
    reg = KPDetector3DMulti('resnet_multi', 18, 64, 3, 15)
    reg.load_state_dict(ora_reg.state_dict())
    phys = PhysiqueMaskGenerator([32, 64, 128])
    phys.load_state_dict(ora_phys.state_dict())
    return reg.cuda().train(), phys.cuda().train(), ora_reg.train(), ora_phys.train()


def _check_wiring(gname, cfg, cams, seed, tol_loss=3e-4, rng_seed=None):
    """Counter3DDisc + Counter3DModel on the HIP path vs a golden written by the imported reference classes
    (tests/golden/make_golden.py: _wiring_case): every loss value, outputs, selected gradients."""
    from modules.model import Counter3DDisc, Counter3DModel
    g = golden(gname)
    reg, phys, _, _ = _hip_models(None, cams)
    disc = gi.seeded_fill_(LinearDisc(), seed=82).cuda()
    gen = Counter3DModel(cfg, reg, None, None, phys)
    dis = Counter3DDisc(cfg, disc, None, None)
    x = {k: T(v).cuda() for k, v in gi.synthetic_batch(2, list(cams), seed=seed).items()}
    if rng_seed is not None:
        torch.manual_seed(rng_seed)
    ld, info = dis(x, gen.regressor)
    assert abs(float(ld) - float(g['loss_disc'])) < 1e-5 + 2e-4 * abs(float(g['loss_disc']))
    ld.mean().backward()
    assert rel(disc.fc.weight.grad, T(g['grad_disc_w'])) < 2e-3 or float(T(g['grad_disc_w']).norm()) == 0
    disc.zero_grad()
    losses, out = gen(x, dis.smpl_discriminator)
    assert sorted('loss_' + k for k in losses) == sorted(k for k in g.files if k.startswith('loss_') and k != 'loss_disc')
    for k, v in losses.items():
        ref = float(g['loss_' + k])
        assert abs(float(v.mean()) - ref) < 1e-5 + tol_loss * abs(ref), (k, float(v.mean()), ref)
    tot = sum(v.mean() for v in losses.values())
    assert abs(float(tot) - float(g['total'])) < 1e-5 + tol_loss * abs(float(g['total']))
    tot.backward()
    torch.cuda.synchronize()                 # gradients are written on several streams (weight-gradient stream, pass chains)
    c0, cl = 'cam_%d' % cams[0], 'cam_%d' % cams[-1]
    assert maxabs(out['pose_3d_depth_' + c0], T(g['pose_3d_cam_0'])) < 0.2      # mm, |coords| ~ 1e3..1e4
    assert maxabs(out['kp_gt_world'], T(g['kp_gt_world'])) < 0.05
    assert maxabs(out['mask_heatmap_line_' + cl][:, :, ::4, ::4], T(g['mask_line_sub'])) < 2e-4
    p = dict(reg.named_parameters())
    # (gradient slices of the PLANTED-PEAK fixture: a smoke check of the wiring - which loss reaches which network at which
    # scale - not the accuracy instrument.  The reference's own fp32 evaluation of this graph sits ~1e-2 from any other
    # evaluation order, and a float64 run of the reference takes other depth peaks altogether (r04: tried as a yardstick - its
    # loss differs by 3 %), so no data-driven bar exists for it; every parameter gradient of the three networks is held to
    # max(floor, 4 x the reference's fp32-vs-fp64 distance) on well-conditioned fixtures instead: detector_allgrads,
    # physique_allgrads, disc_decouple_allgrads)
    assert rel(p['net.head.features.9.bias'].grad, T(g['g_fin_b'])) < 3e-2
    assert rel(p['net.backbone.conv1.weight'].grad, T(g['g_conv1'])) < 5e-2
    if float(T(g['g_phys_dec4_w']).norm()) > 0:
        assert rel(phys.decoder[4].weight.grad, T(g['g_phys_dec4_w'])) < 3e-2
    for key in ('pose_2d_pred_%s_ori' % c0, 'depth_map_' + c0, 'pose_3d_gt_%s_pseudo' % c0, 'mask_physique_' + c0,
                'pose_2d_pred_%s_pseudo' % cl):
        assert key in out
    for key in ('pose_smpl_2d_' + c0, 'pose_smpl_3d_' + cl, 'smpl_logits_' + c0, 'pred_logits_' + cl):
        assert key in info
    return g, reg, phys


@pytest.mark.parametrize('stage', ['S1', 'S2'])
def test_model_wiring_vs_golden(stage):
    _check_wiring('model_HM36_Multi_Sur' + stage, gi.model_params(stage, cam_ids=(0, 1)), (0, 1), 83)


def _yaml_params(name, cams):
    from xas_amd.synthetic import model_config
    mp = model_config(name)['model_params']          # equal to the YAML (tests/test_configs.py)
    mp['cam_id_list'] = list(cams)
    return mp


def test_model_wiring_weighted_mask_losses():
    """HM36_Multi_SurS1 with NON-zero recons / physique_recons weights and use_dis_map: True: the geodesic-weighted
    mask-loss kernels (xas_mask_loss_fwd/bwd modes 2 and 3) and everything upstream of them (line renderer backward,
    physique net backward) are visible to the golden; with the shipped weight 0.0 they are multiplied away."""
    mp = _yaml_params('HM36_Multi_SurS1', (0, 1))
    assert mp['loss_config']['recons_loss']['use_dis_map'] and mp['loss_config']['physique_recons_loss']['use_dis_map']
    mp['loss_config']['recons_loss']['weight'] = 0.02
    mp['loss_config']['physique_recons_loss']['weight'] = 0.02
    g, reg, phys = _check_wiring('model_HM36_Multi_SurS1_wmask', mp, (0, 1), 83)
    assert float(g['loss_reconstruction']) > 1e-4 and float(g['loss_physique_recons']) > 1e-4
    assert float(T(g['g_phys_dec4_w']).norm()) > 0
    assert rel(phys.encoder[0][0].weight.grad, T(g['g_phys_enc0_w'])) < 3e-2
    assert rel(dict(reg.named_parameters())['net.backbone.layer1.0.conv2.weight'].grad[:8], T(g['g_l1c2'])) < 5e-2


def test_model_wiring_use_aug():
    """use_aug branch (model.py:132-140,249-258): same CPU-generator seed as the golden run -> same rotations."""
    mp = _yaml_params('HM36_Multi_SurS2', (0, 1))
    mp['smpl_disc_params']['use_aug'] = True
    g, _, _ = _check_wiring('model_HM36_Multi_SurS2_aug', mp, (0, 1), 86, rng_seed=1234)
    assert float(T(g['g_disc_after_gen']).norm()) > 0


def test_model_wiring_mpi_five_cameras():
    """MPI_Multi_SurS1 (BASELINE config 4) with its own camera list [0, 2, 4, 7, 8]."""
    cams = (0, 2, 4, 7, 8)
    _check_wiring('model_MPI_Multi_SurS1', _yaml_params('MPI_Multi_SurS1', cams), cams, 84)


def test_model_wiring_synth_s2():
    """HM36_Multi_SynthS2 (BASELINE config 5): S2 losses, smpl_pseudo_img_loss.weight 1.0."""
    _check_wiring('model_HM36_Multi_SynthS2', _yaml_params('HM36_Multi_SynthS2', (0, 1)), (0, 1), 85)


@pytest.mark.parametrize('name,batch', [('MPI_Multi_SurS1', 2), ('HM36_Multi_SynthS2', 3)])
def test_train_step_other_configs(name, batch):
    """Full TrainStep (real GCN discriminator, fused Adam, weight-gradient side stream) on the MPI camera list and on
    the SynthS2 configuration: losses of the first step equal the CPU oracle step on the same batch."""
    from oracle import step as ostep
    from oracle.nets import GCNDecouple
    from xas_amd import engine
    from xas_amd.synthetic import model_config
    cfg = model_config(name)
    mp = cfg['model_params']
    cams = mp['cam_id_list']
    torch.manual_seed(5)
    model, disc, od, odisc = engine.prepare_model(cfg)
    oreg = gi.seeded_fill_(ostep.Regressor(**mp['detector_params']), seed=61)
    with torch.no_grad():
        oreg.net.head.features[9].bias.copy_(T(gi.planted_depth_bias(18, 64, seed=62)))
    from oracle.nets import PhysiqueNet
    ophys = gi.seeded_fill_(PhysiqueNet([32, 64, 128]), seed=81)
    ogcn = gi.seeded_fill_(GCNDecouple(mp['smpl_disc_params']), seed=9)
    model.regressor.load_state_dict(oreg.state_dict())
    model.physique_network.load_state_dict(ophys.state_dict())
    disc.smpl_discriminator.load_state_dict(ogcn.state_dict())
    model.cuda().train(), disc.cuda().train()
    disc.smpl_discriminator.header.p = 0.0
    ogcn.parent_ids, ogcn.child_ids = disc.parent_ids, disc.child_ids
    step = engine.TrainStep(cfg, model, disc, od, odisc)
    o_det = torch.optim.Adam(list(oreg.parameters()) + list(ophys.parameters()), lr=od.param_groups[0]['lr'], betas=(0.5, 0.999))
    o_disc = torch.optim.Adam(ogcn.parameters(), lr=odisc.param_groups[0]['lr'], betas=(0.5, 0.999))
    xn = gi.synthetic_batch(batch, cams, seed=95)
    ld, lk, tot, _ = step({k: T(v).cuda() for k, v in xn.items()})
    old, olk = ostep.train_step(mp, oreg.train(), ophys.train(), ogcn, o_det, o_disc, {k: T(v) for k, v in xn.items()})
    assert abs(float(ld) - float(old)) < 1e-5 + 3e-4 * abs(float(old)), (float(ld), float(old))
    assert sorted(lk) == sorted(olk)
    for k in olk:
        assert abs(float(lk[k].mean()) - float(olk[k])) < 1e-5 + 3e-4 * abs(float(olk[k])), (k, float(lk[k].mean()), float(olk[k]))
    assert torch.isfinite(od.param_arena).all() and torch.isfinite(odisc.param_arena).all()


def _sync_oracle_to_hip(pairs, hip_opt, ora_opt):
    """Copy parameters and Adam state of the HIP modules into the oracle's, so that the next step starts from the
    SAME state on both sides.  (One Adam step moves every weight by ~lr * sign(g): where g ~ 0 the sign is decided by
    fp32 rounding, and unsynchronised trajectories drift apart chaotically - measured: 3e-5 ... 2e-1 after two steps
    depending on nothing but the summation order inside the batch-norm reductions.)"""
    sd = hip_opt.state_dict()['state']
    hip_params = [p for g in hip_opt.param_groups for p in g['params']]
    index = {id(p): i for i, p in enumerate(hip_params)}
    with torch.no_grad():
        for hip_mod, ora_mod in pairs:
            op = dict(ora_mod.named_parameters())
            for name, p in hip_mod.named_parameters():
                q = op[name]
                q.copy_(p.detach().cpu())
                st = sd[index[id(p)]]
                ost = ora_opt.state[q]
                ost['exp_avg'].copy_(st['exp_avg'].cpu())
                ost['exp_avg_sq'].copy_(st['exp_avg_sq'].cpu())
                assert float(ost['step']) == float(st['step'])


def test_full_train_step_vs_oracle():
    """disc + gen step with the real GCN discriminator and Adam against the CPU oracle step, three consecutive steps.
    After each step the oracle is re-synchronised to the HIP state (parameters + Adam moments), so every step is an
    exact one-step comparison from identical state with non-trivial optimizer state from the second step on."""
    from modules.discriminator import GCNDiscriminatorDecouple
    from modules.model import Counter3DDisc, Counter3DModel
    from oracle import step as ostep
    from oracle.nets import GCNDecouple
    from xas_amd.engine import TrainStep
    from xas_amd.optim import FusedAdam
    cfg = gi.model_params('S2', cam_ids=(0, 1))
    full = {'model_params': cfg, 'train_params': {'lr_kp_detector': 1e-4, 'lr_discriminator': 1e-4}}
    reg, phys, oreg, ophys = _hip_models('S2', (0, 1))
    odisc = gi.seeded_fill_(GCNDecouple(cfg['smpl_disc_params']), seed=9)
    disc = GCNDiscriminatorDecouple(cfg['smpl_disc_params'])
    disc.load_state_dict(odisc.state_dict())
    disc.cuda().train()
    disc.header.p = 0.0                                         # no dropout on either side
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
    for it in range(3):
        ld, lk, tot, _ = step(xg)
        old, olk = ostep.train_step(cfg, oreg, ophys, odisc, o_det, o_disc, xc)
        assert abs(float(ld) - float(old)) < 1e-5 + 3e-4 * abs(float(old)), (it, float(ld), float(old))
        for k in olk:
            # smpl_gen is evaluated with the discriminator AFTER this step's update (one Adam step from identical state)
            tol = 3e-3 if k == 'smpl_gen' else 3e-4
            assert abs(float(lk[k].mean()) - float(olk[k])) < 1e-5 + tol * abs(float(olk[k])), (it, k, float(lk[k].mean()), float(olk[k]))
        _sync_oracle_to_hip([(reg, oreg), (phys, ophys)], opt_det, o_det)
        _sync_oracle_to_hip([(disc, odisc)], opt_disc, o_disc)
    assert torch.isfinite(opt_det.param_arena).all()


def test_dedupe_step_is_bit_identical(monkeypatch):
    """TrainStep(dedupe=True) computes the real-image detector forward once instead of twice; parameters, Adam
    moments, running statistics and batch counters after two steps must equal the default path's bit for bit.
    The default path itself must be reproducible run to run (no floating-point atomics anywhere in the step).
    (The comparison runs with the pseudo images in a pass of their own, as the dedupe path must: joined with the real
    images - the default - the same sums are formed in a different order.)"""
    import modules.model as mm
    monkeypatch.setattr(mm, 'JOIN_PSEUDO', False)
    from modules.discriminator import GCNDiscriminatorDecouple
    from modules.model import Counter3DDisc, Counter3DModel
    from xas_amd.engine import TrainStep
    from xas_amd.optim import FusedAdam
    cfg = gi.model_params('S2', cam_ids=(0, 1))
    full = {'model_params': cfg, 'train_params': {'lr_kp_detector': 1e-4, 'lr_discriminator': 1e-4}}
    xg = {k: T(v).cuda() for k, v in gi.synthetic_batch(2, [0, 1], seed=93).items()}
    states = []
    for dedupe in (False, False, True):
        reg, phys, _, _ = _hip_models('S2', (0, 1))
        disc = gi.seeded_fill_(GCNDiscriminatorDecouple(cfg['smpl_disc_params']), seed=9).cuda().train()
        disc.header.p = 0.0
        gen, dis = Counter3DModel(cfg, reg, None, None, phys), Counter3DDisc(cfg, disc, None, None)
        opt_det = FusedAdam(list(reg.parameters()) + list(phys.parameters()), lr=1e-4, betas=(0.5, 0.999))
        opt_disc = FusedAdam(disc.parameters(), lr=1e-4, betas=(0.5, 0.999))
        step = TrainStep(full, gen, dis, opt_det, opt_disc, dedupe=dedupe)
        losses = []
        for _ in range(2):
            ld, lk, tot, _ = step(xg)
            losses.append((float(ld), float(tot)))
        torch.cuda.synchronize()
        bufs = {k: v.clone() for k, v in reg.state_dict().items() if 'running' in k or 'num_batches' in k}
        states.append((opt_det.param_arena.clone(), opt_disc.param_arena.clone(), bufs, losses))
    a = states[0]
    for b in states[1:]:
        assert a[3] == b[3]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert a[2].keys() == b[2].keys() and len(a[2]) > 50
        for k in a[2]:
            assert torch.equal(a[2][k], b[2][k]), k
    assert int(a[2]['net.backbone.bn1.num_batches_tracked']) == 2 * (2 + 2 + 2)   # disc + real + pseudo passes, 2 cams


def test_step_switches_are_bit_identical(monkeypatch):
    """The step-level switches that only change WHERE a value is computed must not change any bit of the result:
    sign-mask batch-norm backward with the skip gradient formed in conv1's data-gradient epilogue (XAS_BN_MASK, default on)
    vs the y-reading form with a materialised residual gradient."""
    from modules.discriminator import GCNDiscriminatorDecouple
    from modules.model import Counter3DDisc, Counter3DModel
    from xas_amd.engine import TrainStep
    from xas_amd.optim import FusedAdam
    cfg = gi.model_params('S2', cam_ids=(0, 1))
    full = {'model_params': cfg, 'train_params': {'lr_kp_detector': 1e-4, 'lr_discriminator': 1e-4}}
    xg = {k: T(v).cuda() for k, v in gi.synthetic_batch(2, [0, 1], seed=94).items()}
    states = []
    for mask in ('1', '0'):
        monkeypatch.setenv('XAS_BN_MASK', mask)
        reg, phys, _, _ = _hip_models('S2', (0, 1))
        disc = gi.seeded_fill_(GCNDiscriminatorDecouple(cfg['smpl_disc_params']), seed=9).cuda().train()
        disc.header.p = 0.0
        gen, dis = Counter3DModel(cfg, reg, None, None, phys), Counter3DDisc(cfg, disc, None, None)
        opt_det = FusedAdam(list(reg.parameters()) + list(phys.parameters()), lr=1e-4, betas=(0.5, 0.999))
        opt_disc = FusedAdam(disc.parameters(), lr=1e-4, betas=(0.5, 0.999))
        step = TrainStep(full, gen, dis, opt_det, opt_disc)
        for _ in range(2):
            step(xg)
        torch.cuda.synchronize()
        states.append((opt_det.param_arena.clone(), opt_disc.param_arena.clone()))
    assert torch.equal(states[0][0], states[1][0]) and torch.equal(states[0][1], states[1][1])


@pytest.mark.parametrize('tag,name,use_bn,self_loop', [('res', 'res_gcn', False, True), ('res_bn', 'res_gcn', True, True),
                                                       ('simple_noloop', 'simple_gcn', False, False)])
def test_gcnconv_discriminator_vs_reference_golden(tag, name, use_bn, self_loop):
    """modules.discriminator.GCNDiscriminator (GCNConv, 1 / bone-length edge weights: discriminator.py:80-139) on the GPU
    against goldens of the reference class imported unchanged (make_golden.py: g_disc_gcn): logits, gradient wrt the
    keypoints (through the edge weights), parameter gradients, running mean of the shared norm."""
    from modules.discriminator import GCNDiscriminator
    from oracle.geometry import skeleton_links
    from test_oracle_disc import _load, check_gcn_disc_sequence
    g = golden('disc_gcn_' + tag)
    cfg = dict(gi.model_params('S2')['smpl_disc_params'], name=name, use_bn=use_bn, use_self_loop=self_loop)
    net = GCNDiscriminator(cfg)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net = _load(net, g, seed=94).cuda()
    net.parent_ids, net.child_ids = skeleton_links(gi.HM36_PARENTS, gi.LINE_SELECT, False, False)
    check_gcn_disc_sequence(net, g, tol_logit=5e-5, tol_grad=2e-4, dev='cuda')


def test_optional_step_switches_agree(monkeypatch):
    """Switches that are OFF by default (measured not to pay) still have to compute the same step: batch-norm backward
    reductions in the data-gradient epilogue (ops_nn.FUSE_DGRAD_BN), bias gradients on the side stream
    (ops_nn.BIAS_ON_SIDE), and the two-pass statistics (FUSE_CONV_STATS off).  Different summation orders: parameters after
    one step agree to 2e-5 of the update size scale (lr = 1e-4)."""
    from modules.discriminator import GCNDiscriminatorDecouple
    from modules.model import Counter3DDisc, Counter3DModel
    from xas_amd import ops_nn
    from xas_amd.engine import TrainStep
    from xas_amd.optim import FusedAdam
    cfg = gi.model_params('S2', cam_ids=(0, 1))
    full = {'model_params': cfg, 'train_params': {'lr_kp_detector': 1e-4, 'lr_discriminator': 1e-4}}
    xg = {k: T(v).cuda() for k, v in gi.synthetic_batch(2, [0, 1], seed=95).items()}

    def run(**flags):
        for k, v in flags.items():
            monkeypatch.setattr(ops_nn, k, v)
        reg, phys, _, _ = _hip_models('S2', (0, 1))
        disc = gi.seeded_fill_(GCNDiscriminatorDecouple(cfg['smpl_disc_params']), seed=9).cuda().train()
        disc.header.p = 0.0
        gen, dis = Counter3DModel(cfg, reg, None, None, phys), Counter3DDisc(cfg, disc, None, None)
        opt_det = FusedAdam(list(reg.parameters()) + list(phys.parameters()), lr=1e-4, betas=(0.5, 0.999))
        opt_disc = FusedAdam(disc.parameters(), lr=1e-4, betas=(0.5, 0.999))
        p0 = opt_det.param_arena.clone()
        step = TrainStep(full, gen, dis, opt_det, opt_disc)
        _, _, total, _ = step(xg)
        torch.cuda.synchronize()
        grads_seen = (opt_det.param_arena - p0)
        return float(total.detach()), opt_det.param_arena.clone(), grads_seen

    base = run(FUSE_DGRAD_BN=False, BIAS_ON_SIDE=False, FUSE_CONV_STATS=True)
    for flags in (dict(FUSE_DGRAD_BN=True, BIAS_ON_SIDE=False, FUSE_CONV_STATS=True),
                  dict(FUSE_DGRAD_BN=False, BIAS_ON_SIDE=True, FUSE_CONV_STATS=True),
                  dict(FUSE_DGRAD_BN=False, BIAS_ON_SIDE=False, FUSE_CONV_STATS=False)):
        other = run(**flags)
        assert abs(other[0] - base[0]) < 1e-5 * abs(base[0]), flags
        # Adam's first step moves every weight by ~lr * sign(g): compare where the gradient is not at noise level
        upd = base[2].abs()
        big = upd > 0.5e-4
        assert float(big.float().mean()) > 0.5
        diff = (other[1] - base[1]).abs()[big]
        assert float((diff > 1e-5).float().mean()) < 5e-3, (flags, float(diff.max()))       # sign flips of noise-level gradients
