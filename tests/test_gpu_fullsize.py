"""BASELINE-size (B = 32) checks through size-independent properties: adjoint identities of the conv family,
linearity, batch-norm moment identities, soft-argmax invariants.  No oracle at this size (it would take minutes):
the properties hold for any correct implementation."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize('cin,cout,h,k,stride,pad', [(256, 1152, 64, 1, 1, 0), (64, 64, 64, 3, 1, 1), (256, 256, 16, 3, 1, 1),
                                                     (128, 128, 64, 3, 2, 1), (1024, 2048, 16, 1, 2, 0), (3, 64, 256, 7, 2, 3)])
def test_conv_adjoint_identities_b32(cin, cout, h, k, stride, pad):
    """<dy, conv(x; w)> == <dgrad(dy; w), x> == <wgrad(x, dy), w> for the detector's layer shapes at B = 32."""
    from xas_amd import layers as L
    g = torch.Generator(device='cuda').manual_seed(cin + cout)
    m = L.Conv2d(cin, cout, k, stride, pad, bias=False).cuda()
    x = torch.randn(32, cin, h, h, device='cuda', generator=g).requires_grad_(True)
    y = m(x)
    dy = torch.randn(y.shape, device='cuda', generator=g)
    y.backward(dy)
    torch.cuda.synchronize()
    lhs = _dot(dy, y)
    assert abs(_dot(x.grad, x) - lhs) <= 2e-4 * abs(lhs) + 1e-2
    assert abs(_dot(m.weight.grad, m.weight) - lhs) <= 2e-4 * abs(lhs) + 1e-2


def test_conv_transpose_adjoint_and_linearity_b32():
    from xas_amd import layers as L
    g = torch.Generator(device='cuda').manual_seed(5)
    m = L.ConvTranspose2d(256, 256, 4, 2, 1).cuda()
    x1 = torch.randn(32, 256, 32, 32, device='cuda', generator=g).requires_grad_(True)
    x2 = torch.randn(32, 256, 32, 32, device='cuda', generator=g)
    y1 = m(x1)
    assert y1.shape == (32, 256, 64, 64)
    dy = torch.randn(y1.shape, device='cuda', generator=g)
    y1.backward(dy)
    lhs = _dot(dy, y1)
    assert abs(_dot(x1.grad, x1) - lhs) <= 2e-4 * abs(lhs) + 1e-2
    assert abs(_dot(m.weight.grad, m.weight) - lhs) <= 2e-4 * abs(lhs) + 1e-2
    with torch.no_grad():
        lin = m(2.5 * x1 + x2) - (2.5 * m(x1) + m(x2))
    assert float(lin.abs().max()) < 5e-4 * float(y1.abs().max())


@pytest.mark.parametrize('c,h,act', [(64, 128, 1), (256, 64, 0), (2048, 8, 1), (32, 256, 2)])
def test_batch_norm_moments_b32(c, h, act):
    """Pre-activation output of training-mode BN has per-channel mean beta and variance gamma^2 * var/(var+eps);
    the input gradient is orthogonal to 1 and to xhat per channel."""
    from xas_amd import layers as L
    g = torch.Generator(device='cuda').manual_seed(c)
    m = L.BatchNorm2d(c, act=0).cuda()
    with torch.no_grad():
        m.weight.copy_(torch.rand(c, device='cuda', generator=g) + 0.5)
        m.bias.copy_(torch.randn(c, device='cuda', generator=g))
    x = (torch.randn(32, c, h, h, device='cuda', generator=g) * 3 + 7).requires_grad_(True)
    y = m(x)
    mean = y.mean(dim=(0, 2, 3))
    var = y.var(dim=(0, 2, 3), unbiased=False)
    assert float((mean - m.bias).abs().max()) < 2e-4
    assert float((var / m.weight ** 2 - 1).abs().max()) < 2e-3
    m2 = L.BatchNorm2d(c, act=act).cuda()
    m2.load_state_dict(m.state_dict())
    x2 = x.detach().clone().requires_grad_(True)
    y2 = m2(x2)
    y2.backward(torch.randn(y2.shape, device='cuda', generator=g))
    gx = x2.grad
    xhat = (x2.detach() - x2.detach().mean(dim=(0, 2, 3), keepdim=True))
    scale = float(gx.abs().mean()) * gx[:, 0].numel() + 1e-12
    assert float(gx.sum(dim=(0, 2, 3)).abs().max()) < 2e-3 * scale
    assert float((gx * xhat).sum(dim=(0, 2, 3)).abs().max()) < 2e-3 * scale * float(xhat.abs().mean())


def test_head_invariants_b32():
    """B = 32, K = 18, D = 64 (604 MB of logits): joints inside [-1, 1), x/y shared by the hypotheses, peak bins
    inside [1, D-2] and distinct, gradient of a softmax-based head sums to zero per joint, shifting a joint's
    logits by a constant changes nothing."""
    from xas_amd import ops_head
    g = torch.Generator(device='cuda').manual_seed(1)
    lg = (torch.randn(32, 64, 64, 1152, device='cuda', generator=g) * 2).permute(0, 3, 1, 2).requires_grad_(True)
    kps, dmap, idx = ops_head.softargmax_multi(lg, 18, 3, 15)
    assert kps.shape == (32, 3, 18, 3) and idx.shape == (32, 18, 3) and idx.dtype == torch.int64
    assert float(kps.min()) >= -1.0 and float(kps.max()) < 1.0
    assert torch.equal(kps[:, 0, :, :2], kps[:, 1, :, :2]) and torch.equal(kps[:, 0, :, :2], kps[:, 2, :, :2])
    assert int(idx.min()) >= 1 and int(idx.max()) <= 62
    srt = idx.sort(dim=-1).values
    assert bool((srt[..., 1:] != srt[..., :-1]).all())
    assert abs(float(dmap.sum(dim=1).mean()) - 1.0) < 1e-4
    kps.backward(torch.randn(kps.shape, device='cuda', generator=g))
    gsum = lg.grad.reshape(32, 18, 64, 64, 64).sum(dim=(2, 3, 4))
    assert float(gsum.abs().max()) < 1e-4
    with torch.no_grad():
        shift = torch.randn(32, 18, 1, 1, 1, device='cuda', generator=g).expand(32, 18, 64, 64, 64).reshape(32, 1152, 64, 64)
        k2, _, i2 = ops_head.softargmax_multi(lg.detach() + shift, 18, 3, 15)
    assert float((k2 - kps.detach()).abs().max()) < 1e-4
