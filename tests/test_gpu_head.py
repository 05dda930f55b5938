"""GPU parity: soft-argmax head, patch->world, line-mask renderer (HIP, through the C ABI)
against the CPU oracle and the reference-import goldens."""
import numpy as np
import pytest
import torch

import inputs as gi
from conftest import golden

pytestmark = pytest.mark.gpu

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def dev(a):
    return (T(a) if isinstance(a, np.ndarray) else a).cuda()


def close(a, b, atol, rtol=0.0):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def test_head_small_vs_golden_and_oracle():
    from xas_amd import ops_head
    g = golden('head_small')
    lg = dev(g['logits']).requires_grad_(True)
    kps, dmap, idx = ops_head.softargmax_multi(lg, 2, 3, 15)
    assert idx.dtype == torch.int64 and np.array_equal(idx.cpu().numpy(), g['z_idx'])      # bit exact
    close(kps, g['kps'], 1e-5)                                   # bar is 1e-4 in normalised space
    close(dmap, g['depth_prob_map'], 1e-6)
    (kps * dev(g['grad_out'])).sum().backward()
    close(lg.grad, g['grad_logits'], 2e-7, 2e-4)
    k1, d1 = ops_head.softargmax_single(dev(g['logits']), 2)
    close(k1, g['kps_single'], 1e-5)
    close(d1, g['depth_prob_map_single'], 1e-6)


def test_head_full_size_vs_golden():
    from xas_amd import ops_head
    g = golden('head_full')
    lg, _ = gi.planted_logits(1, 18, 64, seed=12)
    kps, dmap, idx = ops_head.softargmax_multi(dev(lg), 18, 3, 15)
    assert np.array_equal(idx.cpu().numpy(), g['z_idx'])
    close(kps, g['kps'], 2e-5)
    close(dmap, g['depth_prob_map'], 1e-6)
    close(ops_head.softargmax_single(dev(lg), 18)[0], g['kps_single'], 2e-5)


@pytest.mark.parametrize('B,K,D,hy,nb', [(3, 18, 64, 3, 15), (2, 5, 32, 2, 7), (1, 18, 64, 1, 0), (4, 3, 16, 3, 15)])
def test_head_vs_oracle_random(B, K, D, hy, nb):
    from oracle import head as ohead
    from xas_amd import ops_head
    lg_np, _ = gi.planted_logits(B, K, D, seed=100 + B)
    gw = torch.randn(B, hy, K, 3, generator=torch.Generator().manual_seed(1))
    lc = T(lg_np).requires_grad_(True)
    lgpu = dev(lg_np).requires_grad_(True)
    if nb:
        ko, do, io = ohead.softargmax_multi(lc, K, hy, nb)
        kg, dg, ig = ops_head.softargmax_multi(lgpu, K, hy, nb)
        assert np.array_equal(io.numpy(), ig.cpu().numpy())
    else:
        ko, do = ohead.softargmax_single(lc, K)
        kg, dg = ops_head.softargmax_single(lgpu, K)
    close(kg, ko, 2e-5)
    close(dg, do, 1e-6)
    (ko * gw).sum().backward()
    (kg * gw.cuda()).sum().backward()
    close(lgpu.grad, lc.grad, 2e-7, 2e-4)
    # size independent property: softmax gradient sums to zero per (b,k)
    s = lgpu.grad.reshape(B, K, -1).sum(-1).abs().max().item()
    assert s < 1e-5


def test_head_flat_logits_tie_rule():
    """All-equal logits: every inner bin is a 'peak' with equal value -> lowest indices first."""
    from xas_amd import ops_head
    kps, _, idx = ops_head.softargmax_multi(torch.zeros(1, 2 * 16, 16, 16, device='cuda'), 2, 3, 15)
    assert idx.cpu().tolist() == [[[1, 2, 3], [1, 2, 3]]]
    assert torch.isfinite(kps).all()


def test_patch_to_world():
    from oracle import geometry as geo
    from xas_amd import ops_head
    g = golden('geometry')
    cam = gi.camera_params(4, seed=31)
    camg = [dev(a) for a in cam]
    kp = dev(g['kps']).requires_grad_(True)
    w = ops_head.patch_to_world(kp, *camg)
    close(w, g['world'], 1e-2, 3e-6)          # world mm O(1e3..1e4): relative bar
    (w * dev(g['grad_out'])).sum().backward()
    close(kp.grad, g['grad_kps'], 5e-2, 2e-5)
    close(ops_head.patch_to_world(dev(g['kps_px']), *camg, is_norm=False), g['world_px'], 1e-2, 3e-6)
    close(ops_head.patch_to_world(dev(g['kps']), *camg, rect_width=256, mono=True, patch=False), g['world_mono'], 1e-6)
    # all hypotheses at once == per hypothesis
    k4 = torch.randn(4, 3, 18, 3, device='cuda') * 0.5
    w4 = ops_head.patch_to_world(k4, *camg)
    for h in range(3):
        close(w4[:, h], geo.patch_to_world(k4[:, h].cpu(), *[T(a) for a in cam]), 1e-2, 3e-6)
    # mono backward
    km = dev(g['kps']).requires_grad_(True)
    ops_head.patch_to_world(km, *camg, rect_width=256, mono=True, patch=False).sum().backward()
    assert torch.equal(km.grad, -torch.ones_like(km.grad))


@pytest.mark.parametrize('S', [64, 256])
def test_draw_lines_max(S):
    from oracle import geometry as geo
    from xas_amd import ops_head
    p, c = geo.skeleton_links(gi.HM36_PARENTS, gi.LINE_SELECT, False, True)
    g = golden('lines_%d' % S)
    step = 1 if S == 64 else 4
    kp = dev(g['kps']).requires_grad_(True)
    m = ops_head.draw_lines_max(kp, S, p, c, 3.0e-3)
    close(m[:, :, ::step, ::step], g['mask'], 3e-6)
    assert abs(m.double().sum().item() - float(g['checksum'])) < 1e-3 * max(1.0, abs(float(g['checksum'])))
    gw = dev(np.random.Generator(np.random.PCG64(6)).random((2, 1, S, S)).astype(np.float32))
    (m * gw).sum().backward()
    close(kp.grad, g['grad_kps'], 5e-3, 2e-4)


def test_draw_lines_17_and_full_batch():
    from oracle import geometry as geo
    from xas_amd import ops_head
    g = golden('lines_17')
    p17, c17 = geo.skeleton_links(gi.HM36_PARENTS, gi.LINE_SELECT, False, False)
    close(ops_head.draw_lines_max(dev(g['kps']), 64, p17, c17, 3.0e-3), g['mask'], 3e-6)
    # BASELINE size (B=32, 256^2): bounded in [0,1], and identical to the oracle on a slice
    p, c = geo.skeleton_links(gi.HM36_PARENTS, gi.LINE_SELECT, False, True)
    kp = gi.skeleton_2d(32, seed=5)
    m = ops_head.draw_lines_max(dev(kp), 256, p, c, 3.0e-3)
    assert m.shape == (32, 1, 256, 256) and float(m.min()) >= 0 and float(m.max()) <= 1
    close(m[30:32], geo.draw_lines_max(T(kp[30:32]), 256, p, c, 3.0e-3), 3e-6)
    # joints exactly on top of each other (degenerate bone) stay finite
    z = torch.zeros(1, 18, 2, device='cuda', requires_grad=True)
    mz = ops_head.draw_lines_max(z, 64, p, c, 3.0e-3)
    mz.sum().backward()
    assert torch.isfinite(mz).all() and torch.isfinite(z.grad).all()


# ---- the head's first pass in the final convolution's epilogue (forward half; xas_conv_fwd_head) ------------------
def _final_conv(cin, seed, identity=False):
    """layers.Conv2d(cin -> 18 * 64, 1x1, bias) marked as the producer of the head's logits (what KPDetector3DMulti does)."""
    from xas_amd import layers as L
    m = L.Conv2d(cin, 18 * 64, 1, bias=True).cuda()
    m.head_kd = (18, 64)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        if identity:
            m.weight.copy_(torch.eye(18 * 64).view(18 * 64, 18 * 64, 1, 1))
            m.bias.zero_()
        else:
            m.weight.copy_(torch.randn(18 * 64, cin, 1, 1, generator=g) * 0.3)
            m.bias.copy_(torch.randn(18 * 64, generator=g))
    return m


def test_head_in_conv_epilogue_vs_reference_golden():
    """The planted-peak logits of golden `head_full` (written by the reference's KPDetector3DMulti head) pushed through an
    IDENTITY 1x1 convolution whose epilogue emits the head's first-pass records: joints against the reference's (bar 1e-4),
    int64 peak indices bit-exact, depth maps, and the convolution's own output equal to its input to split-arithmetic accuracy."""
    from xas_amd import ops_head, ops_nn
    g = golden('head_full')
    lg, _ = gi.planted_logits(1, 18, 64, seed=12)
    x = dev(lg).contiguous(memory_format=torch.channels_last)
    before = dict(ops_nn.head_stats)
    y = _final_conv(18 * 64, 0, identity=True)(x)
    assert getattr(y, '_xas_head', None) is not None
    kps, dmap, idx = ops_head.softargmax_multi(y, 18, 3, 15)
    assert ops_nn.head_stats['fused'] == before['fused'] + 1 and ops_nn.head_stats['separate'] == before['separate']
    assert float((y - x).abs().max()) < 2e-5 * float(x.abs().max())
    assert np.array_equal(idx.cpu().numpy(), g['z_idx'])
    close(kps, g['kps'], 2e-5)
    close(dmap, g['depth_prob_map'], 1e-6)


@pytest.mark.parametrize('n,cin', [(3, 256), (2, 64), (130, 32)])
def test_head_in_conv_epilogue_equals_the_two_pass_head(n, cin, monkeypatch):
    """Same convolution, same input: records from the epilogue + xas_head_softargmax_from_partials against xas_conv_fwd followed
    by the head's own two passes (XAS_HEAD_IN_EPILOGUE off): logits bit-identical (one kernel, one tile shape), joints to 2e-6
    (64- instead of 128-pixel records: another summation order), peak indices identical, gradients of both forms equal.
    n = 130: more than the 2 GiB / 18.9 MB = 113 images of one launch - the records of the second image range land behind
    the first's."""
    from xas_amd import ops_head, ops_nn
    gen = torch.Generator().manual_seed(n + cin)
    x = (torch.randn(n, cin, 64, 64, generator=gen).cuda() * 0.7).contiguous(memory_format=torch.channels_last)
    conv = _final_conv(cin, 5)
    gw = torch.randn(n, 3, 18, 3, generator=gen).cuda()
    res = []
    for fused in (True, False):
        monkeypatch.setattr(ops_nn, 'HEAD_IN_EPILOGUE', fused)
        before = dict(ops_nn.head_stats)
        xi = x.clone().requires_grad_(True)
        conv.zero_grad()
        y = conv(xi)
        kps, dmap, idx = ops_head.softargmax_multi(y, 18, 3, 15)
        assert ops_nn.head_stats['fused' if fused else 'separate'] == before['fused' if fused else 'separate'] + 1
        (kps * gw).sum().backward()
        torch.cuda.synchronize()
        res.append((y.detach().clone(), kps.detach().clone(), idx.clone(), dmap.clone(), xi.grad.clone(), conv.weight.grad.clone()))
    a, b = res
    assert torch.equal(a[2], b[2])
    assert float((a[1] - b[1]).abs().max()) < 2e-6
    assert float((a[3] - b[3]).abs().max()) < 1e-6
    rel = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm().clamp_min(1e-300))
    assert rel(a[4], b[4]) < 1e-4 and rel(a[5], b[5]) < 1e-4          # (the backward sees kps / statistics that differ by 1e-7)
    if n <= 113:
        assert torch.equal(a[0], b[0])


def test_detector_runs_the_head_from_the_conv_epilogue():
    """KPDetector3DMulti marks its final convolution: one detector forward = one fused launch, no separate first pass."""
    from modules.keypoint_detector_integral_multi import KPDetector3DMulti
    from xas_amd import ops_nn
    det = KPDetector3DMulti('resnet_multi', 18, 64, 3, 15).cuda().train()
    assert det.net.head.features[-1].head_kd == (18, 64)
    before = dict(ops_nn.head_stats)
    kps, _ = det(torch.rand(2, 3, 256, 256).cuda())
    torch.cuda.synchronize()
    assert ops_nn.head_stats['fused'] == before['fused'] + 1 and ops_nn.head_stats['separate'] == before['separate']
    assert kps.shape == (2, 3, 18, 3) and bool(torch.isfinite(kps).all())
