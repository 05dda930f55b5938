"""GPU parity of the layer kernels (through the C ABI) against plain PyTorch fp32 CPU references,
and of the whole detector against the oracle / reference-import goldens."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

import inputs as gi
from conftest import golden

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def maxabs(a, b):
    return float((a.detach().cpu() - b.detach().cpu()).abs().max())


CONV_CASES = [
    # n, cin, h, w, cout, k, stride, pad
    (2, 64, 16, 16, 64, 1, 1, 0),
    (2, 64, 16, 16, 64, 3, 1, 1),
    (2, 128, 17, 13, 96, 3, 2, 1),      # odd sizes, N tile tail
    (3, 256, 8, 8, 512, 1, 2, 0),       # strided 1x1 (downsample)
    (2, 32, 12, 12, 32, 3, 1, 1),       # BN=32 tile
    (2, 3, 32, 32, 64, 7, 2, 3),        # stem (direct path)
    (2, 1, 16, 16, 32, 3, 1, 1),        # physique first conv
    (2, 32, 16, 16, 1, 3, 1, 1),        # physique last conv
    (1, 256, 8, 8, 1152, 1, 1, 0),      # final projection
]


@pytest.mark.parametrize('n,cin,h,w,cout,k,stride,pad', CONV_CASES)
def test_conv2d_fwd_bwd(n, cin, h, w, cout, k, stride, pad):
    from xas_amd import layers as L
    g = torch.Generator().manual_seed(n * 1000 + cin + cout + k)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    xc, wc, bc = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yc = TF.conv2d(xc, wc, bc, stride, pad)
    gy = torch.randn(yc.shape, generator=g)
    (yc * gy).sum().backward()
    m = L.Conv2d(cin, cout, k, stride, pad, bias=True).cuda()
    with torch.no_grad():
        m.weight.copy_(wt)
        m.bias.copy_(b)
    xg = x.cuda().requires_grad_(True)
    yg = m(xg)
    assert yg.shape == yc.shape
    (yg * gy.cuda()).sum().backward()
    assert rel(yg, yc) < 2e-6 and maxabs(yg, yc) < 2e-5
    assert rel(xg.grad, xc.grad) < 2e-6
    assert rel(m.weight.grad, wc.grad) < 3e-6
    assert rel(m.bias.grad, bc.grad) < 3e-6


@pytest.mark.parametrize('tune,what', [(32, 'plain K-loop, buffer loads'), (64 | 128, 'global-load kernels (>= 2 GiB fallback)'),
                                       (64 | 128 | 32, 'global-load kernels, plain loop'), (524288, 'plain weight-gradient loop'),
                                       (8192, 'plain (not XCD-grouped) weight-gradient block order')])
def test_conv_kernel_variants(tune, what):
    """Every conv kernel variant that stays in the library (fallbacks for tensors the 32-bit buffer offsets cannot
    address, the non-pipelined loops, the alternative block order) gives the same results as the shipped ones."""
    from xas_amd import _lib
    try:
        _lib.query('xas_set_precision', _lib.PREC_F32)     # the selectors concern the exact-fp32 kernels
        _lib.query('xas_set_tuning', tune)
        for case in (CONV_CASES[1], CONV_CASES[2], CONV_CASES[3], CONV_CASES[8], (2, 64, 40, 24, 160, 3, 1, 1)):
            test_conv2d_fwd_bwd(*case)
        test_conv_transpose2d(2, 256, 8, 256)
    finally:
        _lib.query('xas_set_tuning', 0)


@pytest.mark.parametrize('h,w', [(64, 64), (50, 70), (256, 256)])
def test_stem_conv_mfma(h, w):
    """7x7 s2 p3, 3 -> 64, no bias: the dedicated MFMA stem kernel (resnet.py:16)."""
    from xas_amd import layers as L
    g = torch.Generator().manual_seed(h)
    x = torch.randn(2, 3, h, w, generator=g)
    m = L.Conv2d(3, 64, 7, 2, 3, bias=False, init='kaiming_fan_out').cuda()
    wc = m.weight.detach().cpu().requires_grad_(True)
    yc = TF.conv2d(x, wc, None, 2, 3)
    gy = torch.randn(yc.shape, generator=g)
    (yc * gy).sum().backward()
    yg = m(x.cuda())
    (yg * gy.cuda()).sum().backward()
    assert yg.shape == yc.shape and rel(yg, yc) < 2e-6 and maxabs(yg, yc) < 2e-5
    assert rel(m.weight.grad, wc.grad) < 3e-6


@pytest.mark.parametrize('n,cin,h,cout', [(2, 2048, 4, 256), (2, 256, 8, 256), (1, 64, 5, 32)])
def test_conv_transpose2d(n, cin, h, cout):
    from xas_amd import layers as L
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, h, h, generator=g)
    wt = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5
    xc, wc = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yc = TF.conv_transpose2d(xc, wc, None, 2, 1)
    gy = torch.randn(yc.shape, generator=g)
    (yc * gy).sum().backward()
    m = L.ConvTranspose2d(cin, cout, 4, 2, 1).cuda()
    with torch.no_grad():
        m.weight.copy_(wt)
    xg = x.cuda().requires_grad_(True)
    yg = m(xg)
    (yg * gy.cuda()).sum().backward()
    assert yg.shape == yc.shape
    assert rel(yg, yc) < 2e-6 and rel(xg.grad, xc.grad) < 2e-6 and rel(m.weight.grad, wc.grad) < 3e-6


@pytest.mark.parametrize('rows,cin,cout', [(576, 6, 128), (576, 128, 128), (32, 4608, 512), (32, 512, 1)])
def test_linear(rows, cin, cout):
    from xas_amd import layers as L
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(rows, cin, generator=g)
    m = L.Linear(cin, cout).cuda()
    xc = x.clone().requires_grad_(True)
    wc, bc = m.weight.detach().cpu().requires_grad_(True), m.bias.detach().cpu().requires_grad_(True)
    yc = TF.linear(xc, wc, bc)
    gy = torch.randn(yc.shape, generator=g)
    (yc * gy).sum().backward()
    xg = x.cuda().requires_grad_(True)
    yg = m(xg)
    (yg * gy.cuda()).sum().backward()
    assert rel(yg, yc) < 2e-6 and rel(xg.grad, xc.grad) < 2e-6
    assert rel(m.weight.grad, wc.grad) < 3e-6 and rel(m.bias.grad, bc.grad) < 3e-6


@pytest.mark.parametrize('lean', [False, True], ids=['shipped', 'lean_reduce'])
@pytest.mark.parametrize('n,c,h,act,res', [(4, 64, 16, 1, False), (2, 256, 9, 1, True), (3, 32, 8, 2, False),
                                           (2, 2048, 4, 0, False), (2, 128, 7, 0, True)])
def test_batch_norm_train(n, c, h, act, res, lean, request):
    """lean: the <= 64-register build of the backward sums (tuning bit 18; shipped in r04, kept as a variant)."""
    from xas_amd import layers as L
    from xas_amd._lib import query
    query('xas_set_tuning', (1 << 18) if lean else 0)
    request.addfinalizer(lambda: query('xas_set_tuning', 0))
    g = torch.Generator().manual_seed(c + h)
    x = torch.randn(n, c, h, h, generator=g) * 2 + 3            # mean >> 0 exercises the pivoted variance
    r = torch.randn(n, c, h, h, generator=g) if res else None
    gam, bet = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    xc = x.clone().requires_grad_(True)
    rc = r.clone().requires_grad_(True) if res else None
    gc, bc = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    yc = TF.batch_norm(xc, rm, rv, gc, bc, True, 0.1, 1e-5)
    if res:
        yc = yc + rc
    yc = {0: lambda t: t, 1: TF.relu, 2: lambda t: TF.leaky_relu(t, 0.01)}[act](yc)
    gy = torch.randn(yc.shape, generator=g)
    (yc * gy).sum().backward()
    m = L.BatchNorm2d(c, act=act).cuda()
    with torch.no_grad():
        m.weight.copy_(gam)
        m.bias.copy_(bet)
    xg = x.cuda().requires_grad_(True)
    rg = r.cuda().requires_grad_(True) if res else None
    yg = m(xg, residual=rg)
    (yg * gy.cuda()).sum().backward()
    assert maxabs(yg, yc) < 2e-5
    assert rel(xg.grad, xc.grad) < 2e-5
    assert rel(m.weight.grad, gc.grad) < 1e-5 and rel(m.bias.grad, bc.grad) < 1e-5
    if res:
        assert rel(rg.grad, rc.grad) < 1e-6
    assert maxabs(m.running_mean, rm) < 1e-6 and maxabs(m.running_var, rv) < 1e-5
    assert int(m.num_batches_tracked) == 1
    # eval mode uses the running statistics
    m.eval()
    ye = m(x.cuda(), residual=r.cuda() if res else None)
    yr = TF.batch_norm(x, rm, rv, gam, bet, False, 0.1, 1e-5)
    if res:
        yr = yr + r
    yr = {0: lambda t: t, 1: TF.relu, 2: lambda t: TF.leaky_relu(t, 0.01)}[act](yr)
    assert maxabs(ye, yr) < 2e-5


def test_pool_upsample_sigmoid():
    from xas_amd import ops_nn as F
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 18, 14, generator=g)
    xc = x.clone().requires_grad_(True)
    yc = TF.max_pool2d(xc, 3, 2, 1)
    gy = torch.randn(yc.shape, generator=g)
    (yc * gy).sum().backward()
    xg = x.cuda().requires_grad_(True)
    yg = F.maxpool3x3s2(xg)
    (yg * gy.cuda()).sum().backward()
    assert maxabs(yg, yc) == 0 and maxabs(xg.grad, xc.grad) < 1e-6
    x = torch.randn(2, 32, 7, 9, generator=g)
    xc = x.clone().requires_grad_(True)
    yc = TF.interpolate(xc, scale_factor=2, mode='bilinear')
    gy = torch.randn(yc.shape, generator=g)
    (yc * gy).sum().backward()
    xg = x.cuda().requires_grad_(True)
    yg = F.upsample2x(xg)
    (yg * gy.cuda()).sum().backward()
    assert maxabs(yg, yc) < 1e-6 and maxabs(xg.grad, xc.grad) < 1e-5
    x = torch.randn(2, 1, 16, 16, generator=g)
    xc = x.clone().requires_grad_(True)
    yc = torch.sigmoid(xc)
    (yc * yc).sum().backward()
    xg = x.cuda().requires_grad_(True)
    yg = F.sigmoid(xg)
    (yg * yg).sum().backward()
    assert maxabs(yg, yc) < 1e-6 and maxabs(xg.grad, xc.grad) < 1e-6


def _hip_regressor(multi=True):
    from modules.keypoint_detector_integral import KPDetector3D
    from modules.keypoint_detector_integral_multi import KPDetector3DMulti
    from oracle import step as ostep
    ora = ostep.Regressor('resnet_multi', 18, 64, 3, 15) if multi else ostep.Regressor('resnet', 18, 64)
    gi.seeded_fill_(ora, seed=61)
    with torch.no_grad():
        ora.net.head.features[9].bias.copy_(T(gi.planted_depth_bias(18, 64, seed=62)))
    hip = KPDetector3DMulti('resnet_multi', 18, 64, 3, 15) if multi else KPDetector3D('resnet', 18, 64)
    hip.load_state_dict(ora.state_dict(), strict=True)            # same key names / shapes as the reference
    return hip.cuda(), ora


def test_detector_vs_golden_and_oracle():
    g = golden('detector')
    hip, ora = _hip_regressor(True)
    assert list(hip.state_dict().keys()) == g['keys'].tolist()
    hip.train()
    x = T(gi.synthetic_batch(2, [0], seed=63)['cam_0_img'])
    kps, dmap = hip(x.cuda())
    assert kps.shape == (2, 3, 18, 3) and dmap.shape == (18, 64)
    assert maxabs(kps, T(g['kps'])) < 1e-4                      # north-star bar: 1e-4 in normalised space
    assert maxabs(dmap, T(g['depth_prob_map'])) < 1e-5
    (kps * T(g['grad_out']).cuda()).sum().backward()
    p = dict(hip.named_parameters())
    # Parameter gradients of this planted-peak case are ill conditioned in fp32: the REFERENCE's own fp32 evaluation sits
    # DEV (relative) from its float64 evaluation of the same graph on these tensors (measured with the imported reference,
    # r03: ReLU / max-pool decisions on near-ties differ between precisions), and any independent fp32-accurate
    # evaluation is as far from float64 again.  Bar = the rule of test_detector_all_parameter_gradients:
    # max(3e-3, 4 x DEV) per tensor (r01-r02 used a flat 3e-2 for all of them; the 64-element slice g_l2ds reads 3.35e-2
    # with bf16x6 products - its bar is now 4.2e-2, g_fin_b's went from 3e-2 to 3e-3).  The per-layer kernels are held to 3e-6 where
    # the comparison is well conditioned (test_every_layer_shape_all_passes_vs_float64).
    DEV = {'g_conv1': 1.07e-2, 'g_l1c2': 9.0e-3, 'g_l2ds': 1.04e-2, 'g_dc0': 4.4e-3, 'g_fin_b': 3.5e-5, 'g_bn1_w': 8.5e-3,
           'norms': 1.0e-2}
    GT = lambda k: max(3e-3, 4.0 * DEV[k])                       # noqa: E731
    assert rel(p['net.backbone.conv1.weight'].grad, T(g['g_conv1'])) < GT('g_conv1')
    assert rel(p['net.backbone.layer1.0.conv2.weight'].grad[:8], T(g['g_l1c2'])) < GT('g_l1c2')
    assert rel(p['net.backbone.layer2.0.downsample.0.weight'].grad[:4, :16], T(g['g_l2ds'])) < GT('g_l2ds')
    assert rel(p['net.head.features.0.weight'].grad[:4, :4], T(g['g_dc0'])) < GT('g_dc0')
    assert rel(p['net.head.features.9.bias'].grad, T(g['g_fin_b'])) < GT('g_fin_b')
    assert rel(p['net.backbone.bn1.weight'].grad, T(g['g_bn1_w'])) < GT('g_bn1_w')
    assert abs(float(p['net.backbone.layer4.2.conv3.weight'].grad.norm()) / float(g['g_l4c3_norm']) - 1) < GT('norms')
    assert abs(float(p['net.head.features.6.weight'].grad.norm()) / float(g['g_dc6_norm']) - 1) < GT('norms')
    sd = hip.state_dict()
    assert maxabs(sd['net.backbone.bn1.running_mean'], T(g['rm_bn1'])) < 1e-6
    assert maxabs(sd['net.backbone.layer3.5.bn3.running_var'], T(g['rv_l3'])) < 1e-5
    # depth-peak indices: bit exact against the oracle on the same weights/input
    ora.train()
    from oracle import head as ohead
    _, _, idx = ohead.softargmax_multi(ora.net(x), 18, 3, 15)
    assert np.array_equal(idx.numpy(), hip.last_peak_indices.cpu().numpy())
    # heat-map (logits) against the golden sub-sample
    hip2, _ = _hip_regressor(True)
    hip2.train()
    heat = hip2.net(x.cuda())
    assert heat.shape == (2, 1152, 64, 64)
    assert maxabs(heat[:, ::37, ::4, ::4], T(g['heat_sub'])) < 5e-4


def test_detector_single_hypothesis():
    hip, _ = _hip_regressor(False)
    hip.train()
    x = T(gi.synthetic_batch(2, [0], seed=63)['cam_0_img'])
    k1, _ = hip(x.cuda())
    assert k1.shape == (2, 1, 18, 3)
    assert maxabs(k1, T(golden('detector_single')['kps'])) < 1e-4


def test_conv_random_shapes():
    """Randomised sweep of the MFMA conv path (buffer-load kernels): channel counts in multiples of 32 and a few that
    are not, odd spatial sizes, strides 1-2, pads 0-2, kernel 1-4 - forward, data gradient and weight gradient against
    torch on the CPU.  Catches tile-tail / padding / tap-mask indexing errors that the fixed layer shapes never hit."""
    import random
    rng = random.Random(20251003)
    done = 0
    while done < 28:
        n = rng.choice([1, 2, 3])
        cin = rng.choice([32, 64, 96, 160, 16, 48])
        cout = rng.choice([32, 64, 96, 128, 192, 16, 40])
        k = rng.choice([1, 2, 3, 4])
        stride = rng.choice([1, 2])
        pad = rng.choice([0, 1, 2])
        h, w = rng.randint(5, 23), rng.randint(5, 23)
        if pad >= k or (h + 2 * pad - k) < 0 or (w + 2 * pad - k) < 0:
            continue
        test_conv2d_fwd_bwd(n, cin, h, w, cout, k, stride, pad)
        done += 1


def test_conv_transpose_random_shapes():
    import random
    rng = random.Random(7)
    for _ in range(8):
        test_conv_transpose2d(rng.choice([1, 2]), rng.choice([32, 64, 128]), rng.randint(3, 11), rng.choice([32, 64, 96]))


@pytest.mark.parametrize('n,c,h,w', [(1, 4, 1, 1), (2, 32, 7, 9), (3, 8, 2, 130), (2, 32, 64, 64), (1, 260, 5, 3), (2, 16, 1, 40)])
def test_upsample2x_shapes(n, c, h, w):
    """x2 bilinear upsampling (physique_network.py:31) and its adjoint against torch on the CPU: rows shorter and longer
    than one 256-lane block, one-pixel borders, channel counts that are not powers of two."""
    from xas_amd import ops_nn as F
    g = torch.Generator().manual_seed(n + c + h + w)
    x = torch.randn(n, c, h, w, generator=g)
    xc = x.clone().requires_grad_(True)
    yc = TF.interpolate(xc, scale_factor=2, mode='bilinear')
    gy = torch.randn(yc.shape, generator=g)
    (yc * gy).sum().backward()
    xg = x.cuda().requires_grad_(True)
    yg = F.upsample2x(xg)
    (yg * gy.cuda()).sum().backward()
    assert yg.shape == yc.shape
    assert maxabs(yg, yc) < 1e-6 and maxabs(xg.grad, xc.grad) < 2e-6


@pytest.mark.parametrize('depth', [18, 34])
def test_basic_block_detectors_vs_oracle(depth):
    """ResNet-18 / 34 detectors (`num_layers` of both detector classes, resnet.py:5-6: torchvision BasicBlock): same state-dict
    keys as the oracle's restatement, joints within 1e-4, parameter gradients of the first and last block within 2e-3."""
    from modules.keypoint_detector_integral_multi import KPDetector3DMulti
    from oracle import step as ostep
    import inputs as gi
    ora = gi.seeded_fill_(ostep.Regressor('resnet_multi', 18, 64, 3, 15, num_layers=depth), seed=40 + depth)
    hip = KPDetector3DMulti('resnet_multi', 18, 64, 3, 15, num_layers=depth)
    assert list(hip.state_dict().keys()) == list(ora.state_dict().keys())
    hip.load_state_dict(ora.state_dict(), strict=True)
    hip.cuda().train(); ora.train()
    x = torch.from_numpy(gi.synthetic_batch(2, [0], seed=7)['cam_0_img'])
    kc, _ = ora(x)
    kg, _ = hip(x.cuda())
    assert float((kg.cpu() - kc).abs().max()) < 1e-4
    gw = torch.randn(kc.shape, generator=torch.Generator().manual_seed(3))
    (kc * gw).sum().backward()
    (kg * gw.cuda()).sum().backward()
    pc, pg = dict(ora.named_parameters()), dict(hip.named_parameters())
    rel = lambda a, b: float((a.detach().cpu().double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    for n in ('net.backbone.layer1.0.conv1.weight', 'net.backbone.layer2.0.downsample.0.weight', 'net.backbone.layer4.1.conv2.weight',
              'net.head.features.0.weight'):
        assert rel(pg[n].grad, pc[n].grad) < 5e-3, (n, rel(pg[n].grad, pc[n].grad))
