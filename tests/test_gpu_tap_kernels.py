"""The kernels that only large problems select - the tap re-use kernels of the stride-1 3x3 layers (igemm_x6t_kernel,
wgrad_x6t_kernel: 8 x 16 pixel patches staged once for the nine taps) and the 64 x 256 tiles of the wide layers - against a
float64 convolution, at sizes where the dispatch takes them (more than 256 tiles of 128 rows), in all three passes.
(The layer-shape sweep of test_gpu_parity_r3.py runs at 2 images, where the small-problem tiles are chosen.)"""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# (n, cin, h, w, cout, k): stride 1, pad k // 2
CASES = [
    (16, 64, 64, 64, 64, 3),       # igemm_x6t<64>, wgrad_x6t<64>: layer1 conv2
    (64, 32, 32, 32, 128, 3),      # igemm_x6t<128> forward, <32> data gradient
    (4, 32, 128, 128, 32, 3),      # igemm_x6t<32>, wgrad_x6t<32>: physique net
    (160, 32, 8, 8, 512, 3),       # 8-pixel-wide maps: patches of two images; 64 x 256 is NOT taken (3x3 goes to the tap kernel)
    (161, 32, 8, 8, 512, 3),       # odd image count on 8-wide maps: falls back to the implicit GEMM
    (16, 64, 64, 64, 256, 1),      # 64 x 256 tiles forward (Cout = 256); data gradient 128 x 64
    (16, 256, 64, 64, 64, 1),      # data gradient with 64 x 256 tiles (Cin = 256)
    (12, 64, 40, 48, 64, 3),       # H = 40, W = 48: five patch rows, three patch columns per image
]


@pytest.mark.parametrize('n,cin,h,w,cout,k', CASES)
def test_large_problem_kernels_vs_float64(n, cin, h, w, cout, k):
    from xas_amd import layers as L
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    g = torch.Generator().manual_seed(n + cin + h + cout + k)
    x = torch.randn(n, cin, h, w, generator=g) * 1.5 + 0.3
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    xc, wc = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yc = TF.conv2d(xc, wc, None, 1, k // 2)
    gy = torch.randn(yc.shape, generator=g)
    (yc * gy.double()).sum().backward()
    m = L.Conv2d(cin, cout, k, 1, k // 2, bias=False).cuda()
    with torch.no_grad():
        m.weight.copy_(wt)
    xg = x.cuda().requires_grad_(True)
    yg = m(xg)
    (yg * gy.cuda()).sum().backward()
    torch.cuda.synchronize()
    e = (rel(yg, yc), rel(xg.grad, xc.grad), rel(m.weight.grad, wc.grad))
    print('fwd / dgrad / wgrad vs float64: %.1e %.1e %.1e' % e)
    assert max(e) < 3e-6, e


def test_tap_and_wide_tile_kernels_equal_the_implicit_gemm():
    """The same problems with the tap re-use kernels and the 64 x 256 tiles switched off (tune bits 22 / 23): results agree to
    accumulation-order noise; the kernel class the library reports does not change."""
    from xas_amd import layers as L
    from xas_amd._lib import query
    outs = {}
    for tune in (0, (1 << 22) | (1 << 23)):
        query('xas_set_tuning', tune)
        try:
            res = []
            for (n, cin, h, w, cout, k) in CASES[:4] + CASES[5:7]:
                g = torch.Generator().manual_seed(7 + cin + cout)
                x = torch.randn(n, cin, h, w, generator=g)
                m = L.Conv2d(cin, cout, k, 1, k // 2, bias=False).cuda()
                with torch.no_grad():
                    m.weight.copy_(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5)
                xg = x.cuda().requires_grad_(True)
                y = m(xg)
                (y * y).sum().backward()
                torch.cuda.synchronize()
                res.append((y.detach().clone(), xg.grad.clone(), m.weight.grad.clone()))
            outs[tune] = res
        finally:
            query('xas_set_tuning', 0)
    for a, b in zip(outs[0], outs[(1 << 22) | (1 << 23)]):
        for ta, tb in zip(a, b):
            assert rel(ta, tb) < 2e-6


def test_batched_weight_preparation_is_bit_identical(monkeypatch):
    """xas_prepare_weights (all layers, one launch, OIHW -> planes) against xas_pack_weight + xas_split_weight per layer."""
    from xas_amd import layers as L
    from xas_amd import ops_nn as F
    torch.manual_seed(3)
    net = torch.nn.Sequential(L.Conv2d(32, 64, 3, 1, 1, bias=False), L.Conv2d(64, 256, 1, 1, 0, bias=False),
                              L.Conv2d(256, 96, 3, 2, 1, bias=False), L.ConvTranspose2d(96, 32, 4, 2, 1)).cuda()
    got = {}
    for batched in (True, False):
        monkeypatch.setattr(F, 'BATCH_PREP', batched)
        for m in net:
            m._cache = F._PackCache()
        if hasattr(net, '_xas_prep'):
            del net._xas_prep
        F.prepack(net)
        torch.cuda.synchronize()
        got[batched] = [{k: v.clone() for k, v in m._cache.packed.items() if k[1]} for m in net]
    n = 0
    for a, b in zip(got[True], got[False]):
        assert a.keys() == b.keys() and a
        for k in a:
            assert torch.equal(a[k].view(torch.uint8), b[k].view(torch.uint8)), k
            n += 1
    assert n >= 8


def test_stem_weight_gradient_kernel_equals_the_general_kernel():
    """stem_wgrad_kernel (LDS patches, tune bit 24 switches it off) against the general weight-gradient kernel and float64."""
    from xas_amd import layers as L
    from xas_amd._lib import query
    g = torch.Generator().manual_seed(11)
    x = torch.randn(6, 3, 64, 96, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    xc, wc = x.double(), wt.double().requires_grad_(True)
    yc = TF.conv2d(xc, wc, None, 2, 3)
    gy = torch.randn(yc.shape, generator=g)
    (yc * gy.double()).sum().backward()
    res = {}
    for tune in (0, 1 << 24):
        query('xas_set_tuning', tune)
        try:
            m = L.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
            with torch.no_grad():
                m.weight.copy_(wt)
            y = m(x.cuda())
            (y * gy.cuda()).sum().backward()
            torch.cuda.synchronize()
            res[tune] = m.weight.grad.clone()
        finally:
            query('xas_set_tuning', 0)
    assert rel(res[0], wc.grad) < 3e-6 and rel(res[1 << 24], wc.grad) < 3e-6
    assert rel(res[0], res[1 << 24]) < 2e-6


def test_stem_forward_f16x3_kernel_against_float64_and_the_fp32_kernel():
    """stem_fwd_f16_kernel (default mode: two fp16 planes of patch and weights, K laid out as 7 filter rows x 24) against a
    float64 convolution and against stem_fwd_kernel (tune bit 25), on an image size with ragged tiles."""
    from xas_amd import layers as L
    from xas_amd._lib import query
    g = torch.Generator().manual_seed(13)
    x = torch.randn(5, 3, 72, 104, generator=g) * 1.5 + 0.2
    wt = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    ref = TF.conv2d(x.double(), wt.double(), None, 2, 3)
    m = L.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
    with torch.no_grad():
        m.weight.copy_(wt)
    res = {}
    for tune in (0, 1 << 25):
        query('xas_set_tuning', tune)
        try:
            with torch.no_grad():
                res[tune] = m(x.cuda()).double().cpu()
        finally:
            query('xas_set_tuning', 0)
    for tune, y in res.items():
        e = float((y - ref).norm() / ref.norm())
        assert e < 1e-6, (tune, e)
    assert float((res[0] - res[1 << 25]).abs().max()) < 2e-5
