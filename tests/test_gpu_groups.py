"""Camera-batched passes: a network run ONCE on the concatenated images of G cameras with grouped batch norm
(ops_nn.bn_groups) must reproduce G separate calls - outputs, gradients, running statistics (in call order) and
batch counters - because that is what the reference does (one detector / physique call per camera, model.py:64,81,147,
231).  Also the SyncBatchNorm merge kernel against torch's formula, and tensors beyond the 2 GiB buffer-offset range."""
import numpy as np
import pytest
import torch

import inputs as gi

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def maxabs(a, b):
    return float((a.detach().cpu() - b.detach().cpu()).abs().max())


@pytest.mark.parametrize('act,residual', [(0, False), (1, False), (2, False), (1, True)])
@pytest.mark.parametrize('n,c,h,w,G', [(2, 64, 8, 8, 3), (4, 256, 5, 7, 2), (32, 64, 16, 16, 4), (3, 1024, 4, 4, 5)])
def test_grouped_batch_norm_equals_separate_calls(n, c, h, w, G, act, residual):
    from xas_amd import layers as L
    from xas_amd import ops_nn as F
    g = torch.Generator().manual_seed(n + c + G + act)
    xs = [(torch.randn(n, c, h, w, generator=g) * (1 + i) + 0.5 * i).cuda() for i in range(G)]
    rs = [torch.randn(n, c, h, w, generator=g).cuda() for _ in range(G)] if residual else [None] * G
    gys = [torch.randn(n, c, h, w, generator=g).cuda() for _ in range(G)]
    def make():
        bn = L.BatchNorm2d(c, act=act).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(0.5 + torch.rand(c, generator=g).cuda())
            bn.bias.copy_(0.3 * torch.randn(c, generator=g).cuda())
        return bn
    a = make()
    b = make()
    b.load_state_dict(a.state_dict())
    # separate calls
    xa = [x.clone().requires_grad_(True) for x in xs]
    ra = [r.clone().requires_grad_(True) if r is not None else None for r in rs]
    ya = [a(x, r) for x, r in zip(xa, ra)]
    sum((y * gy).sum() for y, gy in zip(ya, gys)).backward()
    # one grouped call
    xb = torch.cat(xs).requires_grad_(True)
    rb = torch.cat(rs).requires_grad_(True) if residual else None
    with F.bn_groups(G):
        yb = b(xb, rb)
    (yb * torch.cat(gys)).sum().backward()
    assert maxabs(yb, torch.cat(ya)) < 2e-6 * max(1.0, float(torch.cat(ya).detach().abs().max()))
    assert rel(xb.grad, torch.cat([x.grad for x in xa])) < 3e-6
    if residual:
        assert rel(rb.grad, torch.cat([r.grad for r in ra])) < 3e-6
    assert rel(b.weight.grad, a.weight.grad) < 3e-6 and rel(b.bias.grad, a.bias.grad) < 3e-6
    assert maxabs(b.running_mean, a.running_mean) < 1e-6 and rel(b.running_var, a.running_var) < 1e-6
    assert int(b.num_batches_tracked) == int(a.num_batches_tracked) == G


def test_sync_merge_kernel_vs_torch_formula():
    """xas_bn_sync_merge == torch's batch_norm_gather_stats_with_counts semantics (count-weighted, biased variance),
    including unequal counts and large means (where E[x^2] - mean^2 in fp32 would lose the variance)."""
    from xas_amd._lib import call, ptr
    g = torch.Generator().manual_seed(3)
    world, G, C = 3, 4, 96
    counts = torch.tensor([[100.0, 200, 50, 128]] * world) * torch.tensor([[1.0], [2.0], [0.5]])
    data = [[(torch.randn(int(counts[r, k]), C, generator=g) * (0.01 + k) + 1000.0 * (r + 1)).double() for k in range(G)]
            for r in range(world)]
    stride = 2 * C + 4
    msg = torch.zeros(world, G, stride)
    for r in range(world):
        for k in range(G):
            msg[r, k, :C] = data[r][k].mean(0).float()
            msg[r, k, C:2 * C] = data[r][k].var(0, unbiased=False).float()
            msg[r, k, 2 * C] = counts[r, k]
    mean = torch.empty(G, C, device='cuda')
    var = torch.empty(G, C, device='cuda')
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    mg = msg.cuda()
    call('xas_bn_sync_merge', ptr(mg), world, G, C, stride, ptr(mean), ptr(var), ptr(rm), ptr(rv), 0.1)
    erm, erv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    for k in range(G):
        allk = torch.cat([data[r][k] for r in range(world)])
        # the message carries fp32 per-rank statistics: compare against the exact merge of THOSE numbers
        n = counts[:, k].double()
        m_r, v_r = msg[:, k, :C].double(), msg[:, k, C:2 * C].double()
        em = (m_r * n[:, None]).sum(0) / n.sum()
        ev = ((v_r + (m_r - em) ** 2) * n[:, None]).sum(0) / n.sum()
        assert maxabs(mean[k], em.float()) < 1e-3 * 1e-3 * 3000 and rel(var[k], ev.float()) < 1e-6
        assert rel(var[k], allk.var(0, unbiased=False).float()) < 1e-3          # and close to the true global variance
        tot = float(n.sum())
        erm = 0.9 * erm + 0.1 * em
        erv = 0.9 * erv + 0.1 * ev * tot / (tot - 1)
    assert rel(rm, erm.float()) < 1e-6 and rel(rv, erv.float()) < 1e-5


def _detector(seed=61):
    from modules.keypoint_detector_integral_multi import KPDetector3DMulti
    from oracle import step as ostep
    ora = gi.seeded_fill_(ostep.Regressor('resnet_multi', 18, 64, 3, 15), seed=seed)
    with torch.no_grad():
        ora.net.head.features[9].bias.copy_(T(gi.planted_depth_bias(18, 64, seed=62)))
    det = KPDetector3DMulti('resnet_multi', 18, 64, 3, 15)
    det.load_state_dict(ora.state_dict())
    return det.cuda().train()


def test_detector_forward_groups_equals_separate_calls():
    a, b = _detector(), _detector()
    xn = gi.synthetic_batch(2, [0, 1, 2], seed=71)
    imgs = [T(xn['cam_%d_img' % c]).cuda() for c in (0, 1, 2)]
    gen = torch.Generator().manual_seed(5)
    gws = [torch.randn(2, 3, 18, 3, generator=gen).cuda() for _ in imgs]
    outs = [a(im) for im in imgs]
    sum((k * gw).sum() for (k, _), gw in zip(outs, gws)).backward()
    kb, db = b.forward_groups(torch.cat(imgs), 3)
    (kb * torch.cat(gws)).sum().backward()
    assert db.shape == (3, 18, 64)
    for i, (k, d) in enumerate(outs):
        assert maxabs(kb[2 * i:2 * i + 2], k) < 2e-5
        assert maxabs(db[i], d) < 1e-5
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    # The two evaluations differ only in fp32 summation order, but the parameter gradients of this planted-peak case
    # are ill-conditioned (two fp32 evaluations of the same graph sit ~1e-2 from an fp64 one and
    # from each other in the early layers); the head layers, a few ops from the loss, agree tightly.
    for name, tol in (('net.head.features.9.bias', 1e-4), ('net.head.features.9.weight', 1e-4),
                      ('net.head.features.6.weight', 3e-3), ('net.head.features.3.weight', 5e-3),
                      ('net.backbone.layer4.2.conv3.weight', 1e-2), ('net.backbone.layer3.2.bn2.weight', 3e-2),
                      ('net.backbone.layer2.0.downsample.1.bias', 3e-2), ('net.backbone.layer1.0.conv2.weight', 3e-2),
                      ('net.backbone.conv1.weight', 3e-2)):
        assert rel(pb[name].grad, pa[name].grad) < tol, (name, rel(pb[name].grad, pa[name].grad))
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if 'running' in k:
            assert rel(sb[k], sa[k]) < 1e-5, k
        if 'num_batches' in k:
            assert int(sb[k]) == int(sa[k]) == 3, k


def test_physique_forward_groups_equals_separate_calls():
    from modules.physique_network import PhysiqueMaskGenerator
    from oracle.nets import PhysiqueNet
    ora = gi.seeded_fill_(PhysiqueNet([32, 64, 128]), seed=81)
    a, b = PhysiqueMaskGenerator([32, 64, 128]), PhysiqueMaskGenerator([32, 64, 128])
    a.load_state_dict(ora.state_dict()); b.load_state_dict(ora.state_dict())
    a.cuda().train(); b.cuda().train()
    ms = [T(gi.blob_mask(2, 64, seed=90 + i)).cuda() * 0.9 for i in range(4)]
    xa = [m.clone().requires_grad_(True) for m in ms]
    ya = [a(m) for m in xa]
    sum(y.pow(2).sum() for y in ya).backward()
    xb = torch.cat(ms).requires_grad_(True)
    yb = b.forward_groups(xb, 4)
    yb.pow(2).sum().backward()
    assert maxabs(yb, torch.cat(ya)) < 5e-6
    assert rel(xb.grad, torch.cat([m.grad for m in xa])) < 2e-4
    ga = dict(a.named_parameters())
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if n.endswith('.bias') and n[:-5] + '.weight' in ga and ga[n[:-5] + '.weight'].dim() == 4 and not n.startswith('decoder.4'):
            # the bias of a convolution that feeds a batch norm has NO gradient in exact arithmetic (the norm removes the mean):
            # what both evaluations hold is rounding noise of the column sums, compared against the scale of the layer's
            # weight gradient instead of against itself (the two evaluations split their operands at different maxima - one
            # slot per call - so their noise is not bit-identical)
            scale = float(ga[n[:-5] + '.weight'].grad.norm())
            assert float((pb.grad - pa.grad).norm()) < 2e-4 * scale, n
            continue
        assert rel(pb.grad, pa.grad) < 2e-4, n
    for k, v in a.state_dict().items():
        if 'running' in k:
            assert rel(b.state_dict()[k], v) < 1e-5, k


def test_camera_batched_step_equals_per_camera_step(monkeypatch):
    """Full TrainStep with camera batching on (default) and off: same losses, same parameters after the step (up to
    fp32 summation order)."""
    import modules.model as mm
    from xas_amd import engine
    from xas_amd.synthetic import model_config
    cfg = model_config('HM36_Multi_SurS2')
    cams = [0, 1, 2]
    cfg['model_params']['cam_id_list'] = cams
    xn = gi.synthetic_batch(2, cams, seed=97)
    x = {k: T(v).cuda() for k, v in xn.items()}
    res = []
    for batched, joined in ((False, False), (True, True)):         # default: pseudo images joined into the same pass
        monkeypatch.setattr(mm, 'CAM_BATCH', batched)
        monkeypatch.setattr(mm, 'JOIN_PSEUDO', joined)
        torch.manual_seed(11)
        model, disc, od, odisc = engine.prepare_model(cfg)
        model.cuda().train(), disc.cuda().train()
        disc.smpl_discriminator.header.p = 0.0
        step = engine.TrainStep(cfg, model, disc, od, odisc)
        ld, lk, tot, out = step(x)
        torch.cuda.synchronize()
        res.append((float(ld), {k: float(v.mean()) for k, v in lk.items()}, od.param_arena.clone(), odisc.param_arena.clone(),
                    {k: v.clone() for k, v in model.state_dict().items() if 'running' in k or 'num_batches' in k}, sorted(out)))
    a, b = res
    assert abs(a[0] - b[0]) < 1e-6 + 1e-5 * abs(a[0])
    for k in a[1]:
        assert abs(a[1][k] - b[1][k]) < 1e-6 + 2e-5 * abs(a[1][k]), k
    assert a[5] == b[5]                                        # same output-dict keys
    # Adam's first step moves every weight by ~lr * sign(g): compare the update direction where |g| is not ~0
    assert float((a[2] - b[2]).abs().max()) < 2.5e-4            # 2 * lr at worst (sign flip of a ~zero gradient)
    assert float(((a[2] - b[2]).abs() > 1e-5).float().mean()) < 0.02
    for k in a[4]:
        if 'num_batches' in k:
            assert int(a[4][k]) == int(b[4][k]), k
        else:
            assert rel(b[4][k], a[4][k]) < 1e-4, k


@pytest.mark.parametrize('name,cams,batch', [('HM36_Multi_SurS2', [0, 1, 2], 2), ('HM36_Multi_SurS1', [0, 1, 2, 3], 8)])
def test_joint_prefix_pass_step_equals_separate_passes(name, cams, batch, monkeypatch):
    """r04: the discriminator step's detector pass as a no-grad PREFIX of the generator step's grouped pass (ops_nn prefix
    pass, model.joint_detector_pass; default) against a pass of its own (XAS_JOINT_DISC=0): same losses, same discriminator
    inputs, same parameters and running statistics after the step (up to fp32 summation order), and the backward must not
    have touched the prefix: the batch counters count 3 calls per camera either way."""
    import modules.model as mm
    from xas_amd import engine
    from xas_amd.synthetic import model_config, synthetic_batch
    cfg = model_config(name)
    cfg['model_params']['cam_id_list'] = cams
    x = synthetic_batch(batch, cams, torch.device('cuda'), seed=97)
    res = []
    for joint in (False, True):
        monkeypatch.setattr(mm, 'JOINT_DISC', joint)
        torch.manual_seed(11)
        model, disc, od, odisc = engine.prepare_model(cfg)
        model.cuda().train(), disc.cuda().train()
        disc.smpl_discriminator.header.p = 0.0
        step = engine.TrainStep(cfg, model, disc, od, odisc)
        assert model.joint_pass_possible(x) == joint
        ld, lk, tot, out = step(x)
        torch.cuda.synchronize()
        res.append((float(ld), {k: float(v.mean()) for k, v in lk.items()}, od.param_arena.clone(), odisc.param_arena.clone(),
                    {k: v.clone() for k, v in model.state_dict().items() if 'running' in k or 'num_batches' in k}, sorted(out),
                    {k: v.clone() for k, v in out.items() if isinstance(v, torch.Tensor) and v.is_floating_point()}))
    a, b = res
    assert abs(a[0] - b[0]) < 1e-6 + 1e-5 * abs(a[0]), (a[0], b[0])
    for k in a[1]:
        assert abs(a[1][k] - b[1][k]) < 1e-6 + 2e-5 * abs(a[1][k]), (k, a[1][k], b[1][k])
    assert a[5] == b[5]
    for k in a[6]:
        assert maxabs(a[6][k], b[6][k]) < 1e-4 * max(1.0, float(a[6][k].abs().max())), k
    lr = cfg['train_params']['lr_kp_detector']
    assert float((a[2] - b[2]).abs().max()) < 2.5 * lr          # Adam's first step: 2 * lr at worst (sign flip of a ~zero gradient)
    assert float(((a[2] - b[2]).abs() > 1e-5).float().mean()) < 0.02
    assert float(((a[3] - b[3]).abs() > 1e-5).float().mean()) < 0.02
    for k in a[4]:
        if 'num_batches' in k:
            assert int(a[4][k]) == int(b[4][k]) == (3 * len(cams) if k.startswith('regressor.') else len(cams)), k
        else:
            assert rel(b[4][k], a[4][k]) < 1e-4, k


def test_conv_beyond_2gib_splits_over_images():
    """A gathered operand of >= 2 GiB (the logits gradient of a camera-batched pass) is processed as several launches
    over image ranges: same result as the per-range reference."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr, query
    n, h, w, cin, cout = 36, 64, 64, 64, 3712          # dy: 36 * 4096 * 3712 * 4 B = 2.19 GB
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(n, cin, h, w, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 1, 1, device='cuda', generator=g) * 0.1
    dy = torch.randn(n, cout, h, w, device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    assert dy.numel() * 4 >= 2**31
    shp = F._shape(n, h, w, cin, cout, 1, 1, 1, 0, h, w)
    cache = F._PackCache()
    dx = torch.empty_like(x)
    shp_g, shp_f = F.shape_with_maxima(shp, dy, x), F.shape_with_maxima(shp, x)      # f16x3 kernels in the default mode
    call('xas_conv_dgrad', ptr(dy), ptr(cache.get(wt, 1, shp_g)), ptr(dx), shp_g)
    dw = torch.zeros(cout, cin, 1, 1, device='cuda')
    ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp_g)), device='cuda')
    call('xas_conv_wgrad_acc', ptr(x), ptr(dy), ptr(dw), ptr(ws), shp_g)
    y = torch.empty_like(dy)
    call('xas_conv_fwd', ptr(x), ptr(cache.get(wt, 0, shp_f)), None, ptr(y), shp_f)
    torch.cuda.synchronize()
    # reference per image range with plain matmuls (fp32 on the GPU: an independent code path)
    w2 = wt.view(cout, cin)
    for lo in (0, 17, 35):
        xr = x[lo].permute(1, 2, 0).reshape(-1, cin)
        dyr = dy[lo].permute(1, 2, 0).reshape(-1, cout)
        assert rel(dx[lo].permute(1, 2, 0).reshape(-1, cin), dyr @ w2) < 1e-5
        assert rel(y[lo].permute(1, 2, 0).reshape(-1, cout), xr @ w2.t()) < 1e-5
    ref_dw = torch.zeros(cout, cin, device='cuda', dtype=torch.float64)
    for i in range(n):
        ref_dw += (dy[i].permute(1, 2, 0).reshape(-1, cout).double().t() @ x[i].permute(1, 2, 0).reshape(-1, cin).double())
    assert rel(dw.view(cout, cin), ref_dw.float()) < 1e-5


@pytest.mark.multistream
def test_two_chain_step_equals_single_stream_step(monkeypatch):
    """xas_amd.streams.chains: the real-image pass and the pseudo-image pass of the generator step on two streams (forward
    and backward; running statistics through the bookkeeping stream, norm-parameter gradients by hardware atomics, weight
    gradients through the one side stream) against the default single-stream step with the joined G = 8 pass: same losses,
    same parameters after the step (up to fp32 summation order), same running statistics and counters."""
    import modules.model as mm
    from xas_amd import engine, streams
    from xas_amd.synthetic import model_config
    cfg = model_config('HM36_Multi_SurS2')
    cams = [0, 1, 2]
    cfg['model_params']['cam_id_list'] = cams
    xn = gi.synthetic_batch(2, cams, seed=98)
    x = {k: T(v).cuda() for k, v in xn.items()}
    res = []
    for nchains in (1, 2, 2):
        monkeypatch.setattr(streams, 'CHAINS', nchains)
        torch.manual_seed(12)
        model, disc, od, odisc = engine.prepare_model(cfg)
        model.cuda().train(), disc.cuda().train()
        disc.smpl_discriminator.header.p = 0.0
        step = engine.TrainStep(cfg, model, disc, od, odisc)
        ld, lk, tot, out = step(x)
        torch.cuda.synchronize()
        res.append((float(ld), {k: float(v.mean()) for k, v in lk.items()}, od.param_arena.clone(), odisc.param_arena.clone(),
                    {k: v.clone() for k, v in model.state_dict().items() if 'running' in k or 'num_batches' in k}, sorted(out)))
        step(x)                                            # a second step (re-used streams, re-armed state)
        torch.cuda.synchronize()
        res[-1] += (od.param_arena.clone(),)
    a, b, c = res
    assert torch.equal(b[2], c[2]) and torch.equal(b[3], c[3]) and torch.equal(b[6], c[6])     # deterministic run to run
    assert abs(a[0] - b[0]) < 1e-6 + 1e-4 * abs(a[0])
    for k in a[1]:
        assert abs(a[1][k] - b[1][k]) < 1e-6 + 2e-4 * abs(a[1][k]), k
    assert a[5] == b[5]
    assert float((a[2] - b[2]).abs().max()) < 2.5e-4          # Adam's first step: 2 * lr at worst (sign flip of a ~zero gradient)
    assert float(((a[2] - b[2]).abs() > 1e-5).float().mean()) < 0.02
    for k in a[4]:
        if 'num_batches' in k:
            assert int(a[4][k]) == int(b[4][k]), k
        else:
            assert rel(b[4][k], a[4][k]) < 1e-4, k
