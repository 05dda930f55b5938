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
    call('xas_conv_dgrad_acc', ptr(dyg), ptr(cache.get(wg, 1)), ptr(buf), shp)
    assert rel(buf, ref) < 3e-6
    # and the plain form writes exactly the difference
    out = torch.empty_like(buf)
    call('xas_conv_dgrad', ptr(dyg), ptr(cache.get(wg, 1)), ptr(out), shp)
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
        assert rel(m.bias.grad, bc.grad) < tol, (cin, cout)


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
    call('xas_conv_dgrad_acc_masked', ptr(dyg), ptr(cache.get(wg, 1)), ptr(out), shp, ptr(dpg), ptr(mask))
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
