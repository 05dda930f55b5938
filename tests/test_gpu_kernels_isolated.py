"""Isolated parity tests of C-ABI entry points that the end-to-end tests only see through loose gradient tolerances:
the mask-reconstruction loss in all four modes (loss_func.py:4-16), the accumulating data gradient
(xas_conv_dgrad_acc), the weight gradient accumulated into an existing buffer (xas_conv_wgrad_acc) and the thin
one-channel 3x3 kernels (physique_network.py:41,50), each against the reference-import golden or a plain PyTorch fp32
CPU reference of the same op at <= 3e-6 relative."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from conftest import golden

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


MODES = [('plain', False, False), ('w', True, False), ('clip', False, True), ('w_clip', True, True)]


@pytest.mark.parametrize('tag,use_w,use_clip', MODES)
def test_mask_loss_vs_golden(tag, use_w, use_clip):
    """compute_mask_reconstruction_loss on the HIP path (xas_mask_loss_fwd / _bwd, modes 0-3): value and d/dmask
    against goldens written by the reference's own function (losses.npz: recon_*, grad_*)."""
    from modules.base_losses.loss_func import compute_mask_reconstruction_loss
    g = golden('losses')
    m = T(g['m']).cuda().requires_grad_(True)
    gt, w = T(g['gt']).cuda(), T(g['w']).cuda()
    v = compute_mask_reconstruction_loss(m, gt, weight=w if use_w else None, use_clip=use_clip)
    assert v.dim() == 0
    ref = float(np.asarray(g['recon_' + tag], dtype=np.float64).mean())     # 'clip' golden is the non-scalar tensor
    assert abs(float(v.detach()) - ref) < 1e-7 + 2e-6 * abs(ref), (float(v.detach()), ref)
    (v * 1.7).backward()
    assert rel(m.grad, 1.7 * T(g['grad_' + tag])) < 3e-6
    # the clip threshold is exercised: some pixels fall below 0.1
    assert 0 < int((T(g['m']) <= 0.1).sum()) < g['m'].size


@pytest.mark.parametrize('tag,use_w,use_clip', MODES)
def test_mask_loss_full_size_vs_oracle(tag, use_w, use_clip):
    """Same at the BASELINE size [32,1,256,256] against the oracle (CPU restatement pinned by the golden above)."""
    from modules.base_losses.loss_func import compute_mask_reconstruction_loss
    from oracle import losses as L
    gen = torch.Generator().manual_seed(17)
    m = torch.rand(32, 1, 256, 256, generator=gen)
    gt = (torch.rand(32, 1, 256, 256, generator=gen) > 0.6).float()
    w = 1.0 + 24.0 * torch.rand(32, 1, 256, 256, generator=gen)
    mc = m.clone().requires_grad_(True)
    ref = L.mask_recon(mc, gt, w if use_w else None, use_clip).mean()
    ref.backward()
    mg = m.cuda().requires_grad_(True)
    v = compute_mask_reconstruction_loss(mg, gt.cuda(), weight=w.cuda() if use_w else None, use_clip=use_clip)
    v.backward()
    assert abs(float(v) - float(ref)) < 1e-7 + 3e-6 * abs(float(ref))
    assert rel(mg.grad, mc.grad) < 3e-6


def _conv_case(n, cin, h, w, cout, k, stride, pad, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = torch.randn(n, cout, ho, wo, generator=g)
    return x, wt, dy, ho, wo


# (n, cin, h, w, cout, k, stride, pad): the shapes _Bottleneck.backward sends to xas_conv_dgrad_acc (conv1 of a block:
# 1x1, Cout % 32 == 0, Cin >= 16) plus a 3x3 and a strided case of the same entry point
ACC_CASES = [(2, 256, 16, 16, 64, 1, 1, 0), (2, 1024, 8, 8, 256, 1, 1, 0), (3, 64, 12, 20, 64, 1, 1, 0),
             (2, 128, 9, 11, 96, 3, 1, 1), (2, 64, 16, 16, 128, 3, 2, 1), (1, 2048, 4, 4, 512, 1, 1, 0)]


@pytest.mark.parametrize('n,cin,h,w,cout,k,stride,pad', ACC_CASES)
def test_conv_dgrad_acc(n, cin, h, w, cout, k, stride, pad):
    """dx_buffer += dgrad(dy, W): the buffer already holds the skip-branch gradient (ops_nn._Bottleneck.backward)."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr
    x, wt, dy, ho, wo = _conv_case(n, cin, h, w, cout, k, stride, pad, seed=cin + cout + k)
    g = torch.Generator().manual_seed(1)
    skip = torch.randn(n, cin, h, w, generator=g)
    ref = skip + torch.nn.grad.conv2d_input(x.shape, wt, dy, stride, pad)
    shp = F._shape(n, h, w, cin, cout, k, k, stride, pad, ho, wo)
    assert F._can_accumulate(shp)
    cache = F._PackCache()
    wg = wt.cuda()
    buf = skip.cuda().contiguous(memory_format=torch.channels_last)
    dyg = dy.cuda().contiguous(memory_format=torch.channels_last)
    shp = F.shape_with_maxima(shp, dyg)               # with max |dy|: the f16x3 kernels (without: bf16x6, test_gpu_precision.py)
    call('xas_conv_dgrad_acc', ptr(dyg), ptr(cache.get(wg, 1, shp)), ptr(buf), shp)
    assert rel(buf, ref) < 3e-6
    # and the plain form writes exactly the difference
    out = torch.empty_like(buf)
    call('xas_conv_dgrad', ptr(dyg), ptr(cache.get(wg, 1, shp)), ptr(out), shp)
    assert rel(out, ref - skip) < 3e-6


@pytest.mark.parametrize('n,cin,h,w,cout,k,stride,pad', ACC_CASES + [(2, 32, 24, 24, 32, 3, 1, 1), (2, 256, 8, 8, 256, 4, 2, 1)])
def test_conv_wgrad_acc(n, cin, h, w, cout, k, stride, pad):
    """grad_buffer (OIHW) += wgrad(x, dy): the form the side stream uses to add into the optimizer's gradient arena."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr, query
    x, wt, dy, ho, wo = _conv_case(n, cin, h, w, cout, k, stride, pad, seed=7 + cin + cout + k)
    g = torch.Generator().manual_seed(2)
    prev = torch.randn(cout, cin, k, k, generator=g)
    ref = prev + torch.nn.grad.conv2d_weight(x, wt.shape, dy, stride, pad)
    shp = F._shape(n, h, w, cin, cout, k, k, stride, pad, ho, wo)
    buf = prev.cuda().contiguous()
    ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp)), device='cuda')
    xg = x.cuda().contiguous(memory_format=torch.channels_last)          # named: a temporary could be freed (and its
    dyg = dy.cuda().contiguous(memory_format=torch.channels_last)        # memory re-used) before the launch reads it
    shp = F.shape_with_maxima(shp, dyg, xg)           # both maxima: the f16x3 weight-gradient kernels
    call('xas_conv_wgrad_acc', ptr(xg), ptr(dyg), ptr(buf), ptr(ws), shp)
    assert rel(buf, ref) < 3e-6


@pytest.mark.parametrize('n,c,h,w', [(2, 32, 16, 16), (3, 32, 33, 21), (1, 64, 8, 40), (2, 16, 64, 64), (32, 32, 256, 256)])
def test_thin_one_channel_kernels(n, c, h, w):
    """3x3 s1 p1 convs with one channel on one side: forward, data gradient and weight gradient of both orientations
    (physique_network.py:41 first conv 1 -> C with bias, :50 last conv C -> 1 with bias) vs torch CPU."""
    from xas_amd import layers as L
    g = torch.Generator().manual_seed(n * 100 + c + h)
    for cin, cout in ((1, c), (c, 1)):
        x = torch.randn(n, cin, h, w, generator=g)
        m = L.Conv2d(cin, cout, 3, 1, 1, bias=True).cuda()
        wc = m.weight.detach().cpu().clone().requires_grad_(True)
        bc = m.bias.detach().cpu().clone().requires_grad_(True)
        xc = x.clone().requires_grad_(True)
        yc = TF.conv2d(xc, wc, bc, 1, 1)
        gy = torch.randn(yc.shape, generator=g)
        (yc * gy).sum().backward()
        xg = x.cuda().requires_grad_(True)
        yg = m(xg)
        (yg * gy.cuda()).sum().backward()
        tol = 3e-6 if n * h * w < 100000 else 1e-5        # 2 M-term fp32 sums in the weight gradient at the full size
        assert rel(yg, yc) < 3e-6, (cin, cout)
        assert rel(xg.grad, xc.grad) < 3e-6, (cin, cout)
        assert rel(m.weight.grad, wc.grad) < tol, (cin, cout)
        # (bias gradient: against the float64 sum - the CPU's own fp32 sum of 2 M terms is 1e-5 off, depending on its thread count)
        assert rel(m.bias.grad, gy.double().sum(dim=(0, 2, 3))) < tol, (cin, cout)


@pytest.mark.parametrize('n,cin,h,w,cout,k,stride,pad', ACC_CASES[:4] + [(1, 2048, 4, 4, 512, 1, 1, 0)])
def test_conv_dgrad_acc_masked(n, cin, h, w, cout, k, stride, pad):
    """dx = dgrad(dy, W) + relu'(mask) * dprev with the sign bytes xas_bn_apply writes (block-input gradient of a
    bottleneck without projection; the skip gradient is never materialised)."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr
    x, wt, dy, ho, wo = _conv_case(n, cin, h, w, cout, k, stride, pad, seed=3 + cin + cout + k)
    g = torch.Generator().manual_seed(4)
    dprev = torch.randn(n, cin, h, w, generator=g)
    active = torch.rand(n, cin, h, w, generator=g) > 0.4
    ref = torch.nn.grad.conv2d_input(x.shape, wt, dy, stride, pad) + dprev * active
    # mask bytes over the NHWC float4 order: bit e of byte i = element 4i+e active
    a_nhwc = active.permute(0, 2, 3, 1).reshape(-1, 4).to(torch.uint8)
    mask = (a_nhwc[:, 0] | (a_nhwc[:, 1] << 1) | (a_nhwc[:, 2] << 2) | (a_nhwc[:, 3] << 3)).contiguous().cuda()
    shp = F._shape(n, h, w, cin, cout, k, k, stride, pad, ho, wo)
    cache = F._PackCache()
    wg = wt.cuda()
    dyg = dy.cuda().contiguous(memory_format=torch.channels_last)
    dpg = dprev.cuda().contiguous(memory_format=torch.channels_last)
    out = torch.full_like(dpg, float('nan'))
    shp = F.shape_with_maxima(shp, dyg)
    call('xas_conv_dgrad_acc_masked', ptr(dyg), ptr(cache.get(wg, 1, shp)), ptr(out), shp, ptr(dpg), ptr(mask))
    assert rel(out, ref) < 3e-6


@pytest.mark.parametrize('n,c,h,w,G', [(2, 256, 8, 8, 1), (4, 64, 16, 16, 2), (3, 1024, 4, 4, 3)])
def test_bn_residual_sign_mask_path(n, c, h, w, G, monkeypatch):
    """Batch norm with residual + ReLU: the sign-mask backward (neither pass reads y) against the y-reading form and
    against torch."""
    import torch.nn.functional as TF
    from xas_amd import layers as L
    from xas_amd import ops_nn as F
    g = torch.Generator().manual_seed(c + G)
    x = torch.randn(n * G, c, h, w, generator=g)
    r = torch.randn(n * G, c, h, w, generator=g)
    gy = torch.randn(n * G, c, h, w, generator=g)
    outs = []
    for use_mask in ('1', '0'):
        monkeypatch.setenv('XAS_BN_MASK', use_mask)
        bn = L.BatchNorm2d(c, act=F.ACT_RELU).cuda().train()
        xg, rg = x.cuda().requires_grad_(True), r.cuda().requires_grad_(True)
        with F.bn_groups(G):
            y = bn(xg, rg)
        (y * gy.cuda()).sum().backward()
        outs.append((y.detach(), xg.grad, rg.grad, bn.weight.grad.clone(), bn.bias.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)                          # same arithmetic, different source of the sign
    # torch reference per group
    xs, rs = x.requires_grad_(True), r.requires_grad_(True)
    ys = [torch.relu(TF.batch_norm(xs[i * n:(i + 1) * n], None, None, torch.ones(c), torch.zeros(c), True, 0.1, 1e-5) + rs[i * n:(i + 1) * n])
          for i in range(G)]
    (torch.cat(ys) * gy).sum().backward()
    assert rel(outs[0][0], torch.cat(ys)) < 3e-6 and rel(outs[0][1], xs.grad) < 2e-5 and rel(outs[0][2], rs.grad) < 3e-6


# (n, cin, h, w, cout, k, stride, pad, groups): bottleneck conv/bn pairs.  Covered tiles: 128x128 (Cout >= 96, many rows),
# 64x64 (few blocks), 128x64, 128x32, and shapes whose tile grid does NOT line up with the groups (fallback: second pass)
STATS_CASES = [(8, 64, 32, 32, 256, 1, 1, 0, 1), (8, 64, 32, 32, 256, 1, 1, 0, 4), (16, 256, 64, 64, 64, 1, 1, 0, 8),
               (8, 64, 32, 32, 64, 3, 1, 1, 2), (8, 128, 16, 16, 128, 3, 2, 1, 4), (4, 512, 8, 8, 2048, 1, 1, 0, 2),
               (8, 64, 32, 32, 32, 1, 1, 0, 2), (128, 64, 64, 64, 256, 1, 1, 0, 8),
               (2, 256, 4, 4, 1024, 1, 1, 0, 2), (3, 64, 12, 20, 64, 1, 1, 0, 3), (6, 64, 10, 10, 128, 3, 1, 1, 2),
               (8, 64, 16, 16, 96, 1, 1, 0, 2), (8, 32, 32, 32, 160, 3, 1, 1, 4), (4, 64, 16, 16, 20, 1, 1, 0, 2)]     # ragged column tiles


@pytest.mark.parametrize('n,cin,h,w,cout,k,stride,pad,G', STATS_CASES)
@pytest.mark.parametrize('form', ['local', 'message'])
def test_conv_fwd_bnstats(n, cin, h, w, cout, k, stride, pad, G, form):
    """xas_conv_fwd_bnstats = xas_conv_fwd followed by xas_bn_stats: y bit-identical; mean / biased variance / running
    statistics / SyncBatchNorm message equal to the two-pass kernels (float64 torch statistics of the same y as the judge
    of both), with and without a pivot, whether or not the statistics come out of the conv epilogue."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr, query
    x, wt, _, ho, wo = _conv_case(n, cin, h, w, cout, k, stride, pad, seed=3 + cin + cout + k + G)
    x = x + 0.7                                                          # channel means well away from zero
    shp = F._shape(n, h, w, cin, cout, k, k, stride, pad, ho, wo)
    cache = F._PackCache()
    wg = wt.cuda()
    xg = x.cuda().contiguous(memory_format=torch.channels_last)
    shp = F.shape_with_maxima(shp, xg)
    M, Mg = n * ho * wo, n * ho * wo // G
    y0 = torch.empty(n, cout, ho, wo, device='cuda').contiguous(memory_format=torch.channels_last)
    call('xas_conv_fwd', ptr(xg), ptr(cache.get(wg, 0, shp)), None, ptr(y0), shp)
    rows = y0.permute(0, 2, 3, 1).reshape(G, Mg, cout).double()
    mean64, var64 = rows.mean(1), rows.var(1, unbiased=False)

    gen = torch.Generator().manual_seed(5)
    rm0 = (0.3 * torch.randn(cout, generator=gen)).cuda()
    rv0 = (1.0 + torch.rand(cout, generator=gen)).cuda()
    for pivot in (None, mean64.mean(0).float().contiguous()):
        y = torch.full_like(y0, float('nan'))
        ws = torch.empty(query('xas_conv_fwd_bnstats_workspace_floats', shp, G), device='cuda')
        rm, rv = rm0.clone(), rv0.clone()
        if form == 'local':
            mean = torch.empty(G, cout, device='cuda'); var = torch.empty(G, cout, device='cuda')
            call('xas_conv_fwd_bnstats', ptr(xg), ptr(cache.get(wg, 0, shp)), ptr(y), shp, G, ptr(pivot), ptr(mean), ptr(var),
                 cout, None, ptr(ws), ptr(rm), ptr(rv), 0.1)
            rm_ref, rv_ref = rm0.double(), rv0.double()
            for g in range(G):
                rm_ref = 0.9 * rm_ref + 0.1 * mean64[g]
                rv_ref = 0.9 * rv_ref + 0.1 * var64[g] * (Mg / (Mg - 1))
            assert rel(rm, rm_ref) < 2e-6 and rel(rv, rv_ref) < 2e-6
        else:
            stride_m = 2 * cout + 4
            msg = torch.zeros(G, stride_m, device='cuda')
            call('xas_conv_fwd_bnstats', ptr(xg), ptr(cache.get(wg, 0, shp)), ptr(y), shp, G, ptr(pivot), ptr(msg),
                 ptr(msg[:, cout:]), stride_m, ptr(msg[:, 2 * cout:]), ptr(ws), None, None, 0.1)
            mean, var = msg[:, :cout], msg[:, cout:2 * cout]
            assert torch.equal(msg[:, 2 * cout].cpu(), torch.full((G,), float(Mg)))
            assert torch.equal(rm, rm0) and torch.equal(rv, rv0)
        assert torch.equal(y, y0)
        assert float((mean.double() - mean64).abs().max()) < 2e-6 * float(mean64.abs().max() + var64.max().sqrt())
        assert float(((var.double() - var64).abs() / var64).max()) < 2e-5


# (n, cin, h, w, cout, k, stride, pad, groups): conv3 / conv2 of a bottleneck seen from their data gradient; the norm in
# front has `cin` channels.  Last four: tile grid does not line up (odd rows), strided conv, tiny layer -> three-call fallback
DGRAD_BN_CASES = [(8, 64, 32, 32, 256, 1, 1, 0, 1), (8, 64, 32, 32, 256, 1, 1, 0, 4), (16, 128, 16, 16, 128, 3, 1, 1, 8),
                  (32, 64, 64, 64, 64, 3, 1, 1, 8), (4, 512, 8, 8, 2048, 1, 1, 0, 2), (8, 256, 16, 16, 1024, 1, 1, 0, 2),
                  (3, 64, 12, 20, 64, 1, 1, 0, 3), (4, 128, 16, 16, 128, 3, 2, 1, 2), (2, 256, 4, 4, 1024, 1, 1, 0, 2),
                  (6, 64, 10, 10, 128, 3, 1, 1, 2)]


@pytest.mark.parametrize('n,cin,h,w,cout,k,stride,pad,G', DGRAD_BN_CASES)
def test_conv_dgrad_bn_bwd(n, cin, h, w, cout, k, stride, pad, G):
    """xas_conv_dgrad_bn_bwd = xas_conv_dgrad -> xas_bn_bwd_reduce (ReLU mask from the norm's input) -> xas_bn_bwd_apply:
    gradient wrt the norm's input, the [G][2][C] sums and the in-place parameter-gradient accumulation, against the three
    separate calls and against float64 torch autograd of relu(batch_norm(xb)) -> conv."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr, query
    gen = torch.Generator().manual_seed(11 + cin + cout + k + G)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    xb = torch.randn(n, cin, h, w, generator=gen) * 1.3 + 0.2
    wt = torch.randn(cout, cin, k, k, generator=gen) / (cin * k * k) ** 0.5
    dy = torch.randn(n, cout, ho, wo, generator=gen)
    gamma = 0.5 + torch.rand(cin, generator=gen)
    beta = 0.3 * torch.randn(cin, generator=gen)
    eps = 1e-5
    # float64 reference, group by group
    dx_ref = torch.empty(n, cin, h, w, dtype=torch.float64)
    dgam_ref = torch.zeros(cin, dtype=torch.float64); dbet_ref = torch.zeros(cin, dtype=torch.float64)
    per = n // G
    for g in range(G):
        xg = xb[g * per:(g + 1) * per].double().requires_grad_(True)
        gm, bt = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        hh = torch.relu(TF.batch_norm(xg, None, None, gm, bt, True, 0.1, eps))
        out = TF.conv2d(hh, wt.double(), None, stride, pad)
        (out * dy[g * per:(g + 1) * per].double()).sum().backward()
        dx_ref[g * per:(g + 1) * per] = xg.grad
        dgam_ref += gm.grad; dbet_ref += bt.grad

    shp = F._shape(n, h, w, cin, cout, k, k, stride, pad, ho, wo)
    cache = F._PackCache()
    wg = wt.cuda()
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)
    xbg, dyg = cl(xb), cl(dy)
    M, Mg = n * h * w, n * h * w // G
    rows = xbg.permute(0, 2, 3, 1).reshape(G, Mg, cin)
    mean = rows.mean(1).contiguous(); var = rows.var(1, unbiased=False).contiguous()
    gam, bet = gamma.cuda(), beta.cuda()

    # three separate calls
    dz0 = torch.empty_like(xbg); dx0 = torch.empty_like(xbg)
    call('xas_conv_dgrad', ptr(dyg), ptr(cache.get(wg, 1, shp)), ptr(dz0), shp)
    sums0 = torch.empty(G, 2, cin, device='cuda')
    ws0 = torch.empty(query('xas_bn_workspace_floats', M, cin, G), device='cuda')
    acc_b0 = torch.full((cin,), 0.25, device='cuda'); acc_g0 = torch.full((cin,), -0.5, device='cuda')
    call('xas_bn_bwd_reduce', ptr(xbg), None, ptr(dz0), ptr(mean), ptr(var), ptr(gam), ptr(bet), eps, 1, M, cin, G,
         ptr(sums0), ptr(ws0), ptr(acc_b0), ptr(acc_g0), None)
    call('xas_bn_bwd_apply', ptr(xbg), None, ptr(dz0), ptr(mean), ptr(var), ptr(gam), ptr(bet), ptr(sums0), eps, 1, M, cin,
         G, float(Mg), ptr(dx0), None, None)

    # one call
    dz = torch.full_like(xbg, float('nan')); dx = torch.full_like(xbg, float('nan'))
    sums = torch.empty(G, 2, cin, device='cuda')
    ws = torch.empty(query('xas_conv_dgrad_bn_bwd_workspace_floats', shp, G), device='cuda')
    acc_b = torch.full((cin,), 0.25, device='cuda'); acc_g = torch.full((cin,), -0.5, device='cuda')
    call('xas_conv_dgrad_bn_bwd', ptr(dyg), ptr(cache.get(wg, 1, shp)), shp, ptr(xbg), ptr(mean), ptr(var), ptr(gam), ptr(bet),
         eps, G, float(Mg), ptr(dz), ptr(dx), ptr(sums), ptr(ws), ptr(acc_b), ptr(acc_g))

    scale = float(sums0.abs().max())
    assert float((sums - sums0).abs().max()) < 2e-5 * scale
    assert rel(dx, dx0) < 2e-6
    assert rel(dx, dx_ref) < 5e-6
    assert rel(acc_b - 0.25, dbet_ref) < 2e-5 and rel(acc_g + 0.5, dgam_ref) < 2e-5
    assert rel(acc_b, acc_b0) < 2e-6 and rel(acc_g, acc_g0) < 2e-6
    # without accumulators the sums alone are written
    sums2 = torch.empty_like(sums)
    call('xas_conv_dgrad_bn_bwd', ptr(dyg), ptr(cache.get(wg, 1, shp)), shp, ptr(xbg), ptr(mean), ptr(var), ptr(gam), ptr(bet),
         eps, G, float(Mg), ptr(dz), ptr(dx), ptr(sums2), ptr(ws), None, None)
    assert torch.equal(sums2, sums)


@pytest.mark.parametrize('M,C', [(1000, 32), (8 * 64 * 64, 64), (77, 4), (300000, 128), (5, 260)])
def test_col_sum_acc(M, C):
    """xas_col_sum_acc: acc += column sums (conv bias gradients added into the gradient arena in place)."""
    from xas_amd._lib import call, ptr, query
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g)
    acc0 = torch.randn(C, generator=g)
    xg, acc = x.cuda(), acc0.cuda()
    ws = torch.empty(query('xas_bn_workspace_floats', M, C, 1), device='cuda')
    call('xas_col_sum_acc', ptr(xg), M, C, ptr(acc), ptr(ws))
    ref = acc0.double() + x.double().sum(0)
    assert float((acc.cpu().double() - ref).abs().max()) < 1e-5 * (1 + float(x.abs().sum(0).max()))
    call('xas_col_sum_acc', ptr(xg), M, C, ptr(acc), ptr(ws))                 # twice: accumulates again
    ref = ref + x.double().sum(0)
    assert float((acc.cpu().double() - ref).abs().max()) < 2e-5 * (1 + float(x.abs().sum(0).max()))
