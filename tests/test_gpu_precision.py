"""Arithmetic modes of the MFMA convolutions (xas_hip.h XAS_PREC_*).

The library default is f16x3 (every fp32 operand as two fp16 pieces at a scale taken from the recorded maximum of its
tensor, three partial products; launches that come without those maxima: bf16x6, three bf16 pieces, six partial products;
fp32 accumulation everywhere - both fp32 ACCURATE): every other
GPU test therefore exercises that mode, including every oracle / reference-golden parity test, with tolerances unchanged
from the rounds in which exact-fp32 MFMA was the default.  This file

* re-runs the parity tests of the convolution stack on the EXACT fp32 MFMA kernels (XAS_PREC_F32) and on bf16x6 for every
  pass (XAS_PREC_BF16X6), so all fp32-accurate paths stay pinned ("make it the default and keep an fp32-MFMA
  parametrisation");
* checks that the fp32-accurate modes agree to fp32 rounding;
* checks the plain bf16 variant (XAS_PREC_BF16: operands rounded once, NOT fp32 accurate, reported separately) against an
  fp32 convolution of the bf16-ROUNDED operands (2e-5 relative: summation order only)."""
import pytest
import torch
import torch.nn.functional as TF

from conftest import precision_mode

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


CASES = [(2, 64, 16, 16, 64, 1, 1, 0), (2, 64, 16, 16, 64, 3, 1, 1), (2, 128, 17, 13, 96, 3, 2, 1), (3, 256, 8, 8, 512, 1, 2, 0),
         (2, 32, 12, 12, 32, 3, 1, 1), (1, 256, 8, 8, 1152, 1, 1, 0), (4, 256, 32, 32, 256, 3, 1, 1)]


# ---- the exact-fp32 MFMA kernels under the same parity tests -----------------------------------------------------------
def test_default_precision_is_f16x3():
    from xas_amd import _lib
    import os
    if os.environ.get('XAS_PRECISION', '') == '':
        assert _lib.query('xas_get_precision') == _lib.PREC_F16X3


OTHER_MODES = ['f32', 'bf16x6']        # the fp32-accurate modes that are not the default


@pytest.mark.parametrize('mode', OTHER_MODES)
def test_other_mode_isolated_conv_kernels(mode):
    import test_gpu_nn as T
    with precision_mode(mode):
        for case in T.CONV_CASES:
            T.test_conv2d_fwd_bwd(*case)
        T.test_conv_random_shapes()
        T.test_conv_transpose_random_shapes()


@pytest.mark.parametrize('mode', OTHER_MODES)
def test_other_mode_every_layer_shape(mode):
    import test_gpu_parity_r3 as T
    with precision_mode(mode):
        T.test_every_layer_shape_all_passes_vs_float64()


@pytest.mark.parametrize('mode', OTHER_MODES)
def test_other_mode_detector_goldens(mode):
    import test_gpu_nn as T
    import test_gpu_parity_r3 as R
    with precision_mode(mode):
        T.test_detector_vs_golden_and_oracle()
        T.test_detector_single_hypothesis()
        R.test_detector_all_parameter_gradients()


@pytest.mark.parametrize('mode', OTHER_MODES)
@pytest.mark.parametrize('stage', ['S1', 'S2'])
def test_other_mode_model_wiring(stage, mode):
    import test_gpu_model as T
    with precision_mode(mode):
        T.test_model_wiring_vs_golden(stage)


@pytest.mark.parametrize('mode', OTHER_MODES)
def test_other_mode_full_train_step_vs_oracle(mode):
    import test_gpu_model as T
    with precision_mode(mode):
        T.test_full_train_step_vs_oracle()


@pytest.mark.parametrize('mode', OTHER_MODES)
def test_other_mode_epilogue_variants(mode):
    """accumulating / masked data gradients, statistics epilogue, batch-norm-backward epilogue in the other modes"""
    import test_gpu_kernels_isolated as T
    with precision_mode(mode):
        for case in T.ACC_CASES:
            T.test_conv_dgrad_acc(*case)
            T.test_conv_wgrad_acc(*case)
        for case in T.ACC_CASES[:4]:
            T.test_conv_dgrad_acc_masked(*case)
        for case in T.STATS_CASES:
            for form in ('local', 'message'):
                T.test_conv_fwd_bnstats(*case, form)
        for case in T.DGRAD_BN_CASES:
            T.test_conv_dgrad_bn_bwd(*case)


# ---- the fp32-accurate modes agree --------------------------------------------------------------------------------------
@pytest.mark.parametrize('mode', ['f16x3', 'bf16x6'])
@pytest.mark.parametrize('n,cin,h,w,cout,k,stride,pad', CASES + [(2, 64, 9, 11, 48, 3, 1, 1), (2, 2048, 4, 4, 512, 1, 1, 0)])
def test_conv_split_modes_are_fp32_accurate(n, cin, h, w, cout, k, stride, pad, mode):
    """bf16x6: every operand is split exactly into three bf16 pieces and six exact partial products are accumulated in fp32;
    f16x3 (forward launches): two fp16 pieces (22 bits; weights as 2^10 w), three partial products.  Forward, data gradient
    AND weight gradient must meet the SAME bar as the exact-fp32 MFMA path (3e-6 relative against a float64 convolution of
    the unrounded operands, inputs with a non-zero mean) - three orders of magnitude tighter than plain bf16."""
    with precision_mode(mode):
        _split_mode_case(n, cin, h, w, cout, k, stride, pad)


def test_conv_f16x3_any_operand_magnitude():
    """r04: no fixed activation scale.  Every tensor operand of an f16x3 launch is split at the power-of-two scale its
    recorded maximum selects (ops_nn.act_amax: the producer's slot, else xas_abs_max), so activations of ANY magnitude -
    1e-12 .. 1e8, far outside the old 2^4 window (inf above 2^11, precision loss below 2^-10) - meet the 3e-6 bar of the
    other fp32-accurate modes in all three passes, including a heavy-tailed input (elements down to 1e-6 of the maximum).
    Weights keep their 2^10 scale (|w| < 64, flagged otherwise: test_f16x3_weight_range_is_flagged)."""
    from xas_amd import layers as L
    from xas_amd import ops_nn as O
    from xas_amd._lib import query
    for xs, ws in ((1e-12, 1.0), (1e-5, 1.0), (1e-2, 1.0), (3e2, 1.0), (5e3, 1.0), (1e8, 1.0), (1.0, 1e-3), (1.0, 30.0), (5e3, 1e-2)):
        g = torch.Generator().manual_seed(7)
        x = torch.randn(2, 128, 16, 16, generator=g) * torch.exp(torch.randn(2, 128, 16, 16, generator=g) * 2.0) * xs
        wt = torch.randn(64, 128, 3, 3, generator=g) * ws / 34.0
        gy = torch.randn(2, 64, 16, 16, generator=g)
        m = L.Conv2d(128, 64, 3, 1, 1, bias=False).cuda()
        with torch.no_grad():
            m.weight.copy_(wt)
        with precision_mode('f16x3'):
            xg = x.cuda().requires_grad_(True)
            y = m(xg)
            # dy with its maximum, as a norm's backward would hand it over
            dy = gy.cuda().contiguous(memory_format=torch.channels_last)
            O.tag_grad_amax(dy, O.amax_slot_from_value(dy.abs().max()))
            y.backward(dy)
            torch.cuda.synchronize()
        xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
        yd = TF.conv2d(xd, wd, None, 1, 1)
        yd.backward(gy.double())
        assert torch.isfinite(y).all()
        assert rel(y, yd) < 3e-6, (xs, ws, rel(y, yd))
        assert rel(xg.grad, xd.grad) < 3e-6, (xs, ws)
        assert rel(m.weight.grad, wd.grad) < 3e-6, (xs, ws, rel(m.weight.grad, wd.grad))


def test_f16x3_launch_without_maxima_runs_as_bf16x6():
    """A call that does not come with the maximum of its tensor operand (a maintainer binding the C ABI without the amax
    protocol) never meets fp16's range: it runs on the bf16x6 kernels (kernel class 3, three-plane weights) - |x| = 5e3 and
    1e-5, which the r03 fixed scale turned into inf / lost bits, are exact to the bar; with the maximum the same call is
    class 4 (f16x3)."""
    from xas_amd import ops_nn as O
    from xas_amd._lib import call, ptr, query
    n, cin, h, w, cout = 2, 64, 16, 16, 64
    g = torch.Generator().manual_seed(3)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 24.0
    with precision_mode('f16x3'):
        for xs in (5e3, 1e-5):
            x = torch.randn(n, cin, h, w, generator=g) * xs
            xg = O.to_cl(x.cuda())
            shp = O._shape(n, h, w, cin, cout, 3, 3, 1, 1, h, w)
            assert query('xas_conv_kernel_class', shp, 0) == 3 and query('xas_conv_weight_planes', shp, 0) == 3
            assert query('xas_conv_kernel_class', shp, 1) == 3 and query('xas_conv_kernel_class', shp, 2) == 3
            cache = O._PackCache()
            wg = wt.cuda()
            y = O.empty_cl(n, cout, h, w, xg)
            call('xas_conv_fwd', ptr(xg), ptr(cache.get(wg, 0, shp)), None, ptr(y), shp)
            ref = TF.conv2d(x.double(), wt.double(), None, 1, 1)
            assert torch.isfinite(y).all() and rel(y, ref) < 3e-6, (xs, rel(y, ref))
            shp_m = O.shape_with_maxima(shp, xg)
            assert query('xas_conv_kernel_class', shp_m, 0) == 4 and query('xas_conv_weight_planes', shp_m, 0) == 2
            y2 = O.empty_cl(n, cout, h, w, xg)
            call('xas_conv_fwd', ptr(xg), ptr(cache.get(wg, 0, shp_m)), None, ptr(y2), shp_m)
            assert torch.isfinite(y2).all() and rel(y2, ref) < 3e-6, (xs, rel(y2, ref))
            # a weight gradient needs BOTH maxima for f16x3
            dy = O.to_cl(torch.randn(n, cout, h, w, generator=g).cuda())
            only_dy = O.shape_with_maxima(shp, dy)
            assert query('xas_conv_kernel_class', only_dy, 2) == 3
            both = O.shape_with_maxima(shp, dy, xg)
            assert query('xas_conv_kernel_class', both, 2) == 4


def test_maxima_travel_with_the_tensors_on_the_hot_path():
    """The detector's convolutions find the maximum of their input on the tensor (recorded by the norm / pooling kernel that
    wrote it): one xas_abs_max per pass - for the images - and none in the backward.  A stale tag (a tensor that outlived its
    step: the slot arena has been rewound) is not used."""
    from modules.keypoint_detector_integral_multi import KPDetector3DMulti
    from xas_amd import ops_nn as O
    from xas_amd.synthetic import model_config
    cfg = model_config('HM36_Multi_SurS1')['model_params']['detector_params']
    torch.manual_seed(0)
    det = KPDetector3DMulti(**cfg).cuda().train()
    img = torch.rand(2, 3, 256, 256, device='cuda')
    with precision_mode('f16x3'):
        O.reset_grad_amax()
        before = O.amax_stats['abs_max']
        kps = det(img)[0]
        assert O.amax_stats['abs_max'] - before == 1
        kps.square().sum().backward()
        assert O.amax_stats['abs_max'] - before == 1
        # stale tags: after the arena is rewound the image is measured again
        xin = O.to_cl(torch.randn(2, 64, 8, 8, device='cuda'))
        s1 = O.act_amax(xin)
        assert O.amax_of(xin) is s1
        O.reset_grad_amax()
        assert O.amax_of(xin) is None
        xin.mul_(2.0)
        s2 = O.act_amax(xin)
        torch.cuda.synchronize()
        assert float(s2.max()) == float(xin.abs().max()) and int((s2 != 0).sum()) <= 32
        xin.add_(1.0)                                   # modified in place: the recorded maximum no longer describes it
        assert O.amax_of(xin) is None


def test_f16x3_stem_weight_range_is_flagged():
    """The stem kernel splits its 9 408 weights itself (2^10 w): |w| >= 64 raises the same device flag as the
    weight-preparation kernels (test_f16x3_weight_range_is_checked), read and cleared by xas_f16_weight_overflow."""
    from xas_amd import layers as L
    from xas_amd._lib import query
    with precision_mode('f16x3'), torch.no_grad():
        query('xas_f16_weight_overflow', 1)
        stem = L.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
        img = torch.rand(2, 3, 64, 64, device='cuda')
        y = stem(img)
        ref = TF.conv2d(img.double().cpu(), stem.weight.detach().double().cpu(), None, 2, 3)
        assert rel(y, ref) < 3e-6
        assert query('xas_f16_weight_overflow', 1) == 0
        stem.weight[10, 1, 3, 3] = -65.0
        stem(img)
        assert query('xas_f16_weight_overflow', 1) == 1 and query('xas_f16_weight_overflow', 0) == 0      # read-and-clear


@pytest.mark.parametrize('scale', [1e-9, 1e-5, 1.0, 1e4])
@pytest.mark.parametrize('n,cin,h,w,cout,k,stride,pad', [(2, 64, 16, 16, 64, 3, 1, 1), (3, 256, 8, 8, 512, 1, 2, 0),
                                                         (2, 128, 17, 13, 96, 3, 2, 1), (4, 256, 32, 32, 256, 3, 1, 1),
                                                         (2, 1024, 16, 16, 256, 1, 1, 0)])
def test_f16x3_gradients_with_amax(n, cin, h, w, cout, k, stride, pad, scale):
    """Gradient launches of XAS_PREC_F16X3: dy is split into two fp16 pieces at the power-of-two scale given by max |dy|
    (xas_conv_shape.grad_amax).  Any magnitude of the gradient tensor - 1e-9 .. 1e4 - meets the bar of the other
    fp32-accurate modes, with a heavy-tailed dy (elements down to 1e-6 of the maximum)."""
    from xas_amd import ops_nn as O
    from xas_amd._lib import ConvShape, call, ptr, query
    g = torch.Generator().manual_seed(n + cin + k)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    dy = torch.randn(n, cout, ho, wo, generator=g) * torch.exp(torch.randn(n, cout, ho, wo, generator=g) * 2.0) * scale
    dref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double(), dy.double(), stride, pad)
    with precision_mode('f16x3'):
        dyc = O.to_cl(dy.cuda())
        amax = O.amax_slot_from_value(dyc.abs().max())
        shp = ConvShape(n, h, w, cin, cout, k, k, stride, pad, ho, wo, 0, amax.data_ptr())
        assert query('xas_conv_weight_planes', shp, 1) == 2
        cache = O._PackCache()
        wc = wt.cuda()
        dx = O.empty_cl(n, cin, h, w, dyc)
        call('xas_conv_dgrad', ptr(dyc), ptr(cache.get(wc, 1, shp)), ptr(dx), shp)
        # weight gradient: x and dy each at the scale of its maximum
        x = torch.randn(n, cin, h, w, generator=g) * 2.0 + 0.3
        xc = O.to_cl(x.cuda())
        xmax = O.amax_slot_from_value(xc.abs().max())
        shp = ConvShape(n, h, w, cin, cout, k, k, stride, pad, ho, wo, 0, amax.data_ptr(), xmax.data_ptr())
        assert query('xas_conv_kernel_class', shp, 2) in (4, 1)          # (shapes outside the split kernels: exact fp32)
        dw = torch.empty(cout, cin, k, k, device='cuda')
        ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp)), device='cuda')
        call('xas_conv_wgrad_oihw', ptr(xc), ptr(dyc), ptr(dw), ptr(ws), shp)
        torch.cuda.synchronize()
    assert rel(dx, dref) < 3e-6, rel(dx, dref)
    wref = torch.nn.grad.conv2d_weight(x.double(), wt.shape, dy.double(), stride, pad)
    assert rel(dw, wref) < 3e-6, rel(dw, wref)


@pytest.mark.parametrize('n,cin,h,w,cout,k', [(16, 64, 64, 64, 64, 3), (64, 32, 32, 32, 128, 3), (4, 32, 128, 128, 32, 3),
                                             (160, 32, 8, 8, 512, 3), (16, 64, 64, 64, 256, 1), (16, 256, 64, 64, 64, 1)])
def test_f16x3_gradients_with_amax_large_problem_kernels(n, cin, h, w, cout, k):
    """The same at sizes where the dispatch selects the tap re-use kernels (igemm_x6t / wgrad_x6t) and the 64 x 256 tiles
    (cases of test_gpu_tap_kernels.py), gradient magnitude 1e-4."""
    test_f16x3_gradients_with_amax(n, cin, h, w, cout, k, 1, k // 2, 1e-4)


@pytest.mark.parametrize('scale', [1e-6, 1.0, 1e3])
def test_f16x3_conv_transpose_with_norm(scale):
    """ConvTranspose2d -> BatchNorm -> ReLU (deconv_head.py:27-32) in the default mode: the forward is a data-gradient-type
    launch on an ACTIVATION (fixed scale), its backward a forward-type launch and a weight gradient whose gradient tensor
    is the `x` argument (XAS_GRAD_IS_X), both scaled by the maximum the norm's backward recorded.  Against float64 autograd."""
    from xas_amd import layers as L
    from xas_amd._lib import query
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 64, 8, 8, generator=g) + 0.2
    ct = L.ConvTranspose2d(64, 96, 4, 2, 1).cuda()
    bn = L.BatchNorm2d(96).cuda().train()
    gy = (torch.randn(4, 96, 16, 16, generator=g) * scale)
    xg = x.cuda().requires_grad_(True)
    y = torch.relu(bn(ct(xg)))
    (y * gy.cuda()).sum().backward()
    assert query('xas_get_precision') == 3
    wd = ct.weight.detach().double().cpu().requires_grad_(True)
    xd = x.double().requires_grad_(True)
    gd, bd = bn.weight.detach().double().cpu(), bn.bias.detach().double().cpu()
    yd = torch.relu(TF.batch_norm(TF.conv_transpose2d(xd, wd, None, 2, 1), None, None, gd, bd, True, 0.1, bn.eps))
    (yd * gy.double()).sum().backward()
    assert rel(y, yd) < 3e-6
    assert rel(xg.grad, xd.grad) < 2e-5, rel(xg.grad, xd.grad)            # (through the norm's backward: cancellation)
    assert rel(ct.weight.grad, wd.grad) < 2e-5, rel(ct.weight.grad, wd.grad)


def test_f16x3_gradient_with_two_consumers_falls_back():
    """A conv output with TWO consumers: autograd sums the two gradients (possibly in place into the tensor that carries a
    recorded maximum).  The conv backward must not trust a maximum recorded before the accumulation: result against
    float64 autograd at the usual bar, with the second gradient 1e4 times larger than the first."""
    from xas_amd import layers as L
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 64, 16, 16, generator=g)
    conv = L.Conv2d(64, 64, 3, 1, 1, bias=False).cuda()
    bn = L.BatchNorm2d(64).cuda().train()
    gy = torch.randn(4, 64, 16, 16, generator=g) * 1e-3
    xg = x.cuda().requires_grad_(True)
    y = conv(xg)
    out = (torch.relu(bn(y)) * gy.cuda()).sum() + (y * 10.0 * gy.cuda()).sum()          # norm branch + a direct branch
    out.backward()
    wd = conv.weight.detach().double().cpu().requires_grad_(True)
    xd = x.double().requires_grad_(True)
    yd = TF.conv2d(xd, wd, None, 1, 1)
    od = (torch.relu(TF.batch_norm(yd, None, None, bn.weight.detach().double().cpu(), bn.bias.detach().double().cpu(), True, 0.1, bn.eps))
          * gy.double()).sum() + (yd * 10.0 * gy.double()).sum()
    od.backward()
    assert rel(xg.grad, xd.grad) < 1e-5, rel(xg.grad, xd.grad)
    assert rel(conv.weight.grad, wd.grad) < 1e-5, rel(conv.weight.grad, wd.grad)


def _split_mode_case(n, cin, h, w, cout, k, stride, pad):
    from xas_amd import layers as L
    g = torch.Generator().manual_seed(cin + cout + k + n)
    x = torch.randn(n, cin, h, w, generator=g) * 3.0 + 0.5
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    m = L.Conv2d(cin, cout, k, stride, pad, bias=False).cuda()
    with torch.no_grad():
        m.weight.copy_(wt)
    xg = x.cuda().requires_grad_(True)
    y = m(xg)
    exact = TF.conv2d(x.double(), wt.double(), None, stride, pad)
    assert rel(y, exact) < 3e-6, rel(y, exact)
    gy = torch.randn(exact.shape, generator=g)
    (y * gy.cuda()).sum().backward()
    dref = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), stride, pad)
    assert rel(xg.grad, dref) < 3e-6, rel(xg.grad, dref)
    wref = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double(), stride, pad)
    assert rel(m.weight.grad, wref) < 3e-6, rel(m.weight.grad, wref)
    with precision_mode('f32'):                               # the exact-fp32 MFMA path on the same inputs
        y32 = m(x.cuda())
    assert rel(y, y32) < 2e-6


def _planted_detector():
    import inputs as gi
    from modules.keypoint_detector_integral_multi import KPDetector3DMulti
    from oracle import step as ostep
    ora = gi.seeded_fill_(ostep.Regressor('resnet_multi', 18, 64, 3, 15), seed=61)
    with torch.no_grad():
        ora.net.head.features[9].bias.copy_(torch.from_numpy(gi.planted_depth_bias(18, 64, seed=62)))
    det = KPDetector3DMulti('resnet_multi', 18, 64, 3, 15)
    det.load_state_dict(ora.state_dict())
    return det.cuda().train(), ora


@pytest.mark.parametrize('mode', ['f16x3', 'bf16x6'])
def test_detector_split_modes_match_fp32_path(mode):
    """End to end through 56 conv layers and the soft-argmax head: joints of the split modes against the exact-fp32 MFMA
    path on the same weights and images - within the fp32 path's own distance to the reference golden, far inside 1e-4."""
    import inputs as gi
    det, ora = _planted_detector()
    img = torch.from_numpy(gi.synthetic_batch(2, [0], seed=5)['cam_0_img']).cuda()
    with torch.no_grad():
        with precision_mode(mode):
            k6, _ = det(img)
        det.load_state_dict(ora.state_dict())          # same running statistics for the second pass
        with precision_mode('f32'):
            k32, _ = det(img)
    d = float((k6 - k32).abs().max())
    print(mode, 'vs fp32 MFMA, max |d joint| =', d)
    assert d < 1e-5


# ---- plain bf16 (variant) -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n,cin,h,w,cout,k,stride,pad', CASES)
def test_conv_bf16_all_passes(n, cin, h, w, cout, k, stride, pad):
    from xas_amd import layers as L
    g = torch.Generator().manual_seed(cin + cout + k + n)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    r = lambda t: t.bfloat16().float()                       # round to nearest even, as v_cvt_pk_bf16_f32
    m = L.Conv2d(cin, cout, k, stride, pad, bias=False).cuda()
    with torch.no_grad():
        m.weight.copy_(wt)
    with precision_mode('bf16'):
        xg = x.cuda().requires_grad_(True)
        y = m(xg)
        ref = TF.conv2d(r(x), r(wt), None, stride, pad)
        exact = TF.conv2d(x, wt, None, stride, pad)
        assert rel(y, ref) < 2e-5, rel(y, ref)
        assert rel(y, exact) < 6e-3
        gy = torch.randn(ref.shape, generator=g)
        (y * gy.cuda()).sum().backward()
        torch.cuda.synchronize()
    dref = torch.nn.grad.conv2d_input(x.shape, r(wt), r(gy), stride, pad)
    assert rel(xg.grad, dref) < 2e-5, rel(xg.grad, dref)
    wref = torch.nn.grad.conv2d_weight(r(x), wt.shape, r(gy), stride, pad)
    wexact = torch.nn.grad.conv2d_weight(x, wt.shape, gy, stride, pad)
    # (shapes the bf16-split weight-gradient kernel does not take - here the 4 x 4 output grid - run the exact-fp32 kernel)
    assert rel(m.weight.grad, wref) < 2e-5 or rel(m.weight.grad, wexact) < 3e-6, (rel(m.weight.grad, wref), rel(m.weight.grad, wexact))


def test_conv_transpose_bf16():
    from xas_amd import layers as L
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 256, 8, 8, generator=g)
    wt = torch.randn(256, 256, 4, 4, generator=g) / 32
    r = lambda t: t.bfloat16().float()
    m = L.ConvTranspose2d(256, 256, 4, 2, 1).cuda()
    with torch.no_grad():
        m.weight.copy_(wt)
    with precision_mode('bf16'):
        y = m(x.cuda())
    assert rel(y, TF.conv_transpose2d(r(x), r(wt), None, 2, 1)) < 2e-5


def test_detector_bf16_close_to_fp32():
    """End to end: predicted joints of the bf16 variant vs the fp32-accurate path on the same weights and images (reported,
    not a parity claim: the 1e-4 bar belongs to the fp32-accurate modes)."""
    import inputs as gi
    det, _ = _planted_detector()
    x = torch.from_numpy(gi.synthetic_batch(2, [0], seed=63)['cam_0_img']).cuda()
    with torch.no_grad():
        with precision_mode('bf16'):
            kb, _ = det(x)
        kf, _ = det(x)
    diff = (kb - kf).abs()
    d = float(diff.max())
    print('bf16 vs fp32-accurate detector, max |kps diff| = %.3e (normalised patch units)' % d)
    # once-rounded operands can swap the order of two near-equal depth peaks of a joint (a jump of O(1), seen when the
    # summation order inside the kernel changes); everything else stays within a few 1e-3
    assert float((diff < 5e-2).float().mean()) >= 0.9 and float(diff.median()) < 1e-2


def test_free_running_steps_split_modes_vs_exact_fp32():
    """Three consecutive optimisation steps (disc + gen, Adam, no re-synchronisation) in the default mode and in bf16x6
    against the exact-fp32 kernels from the same initial state and batch.  Step 1 (identical state): every loss term within
    1e-5 relative.  Step 2 (after one Adam update, which maps ANY non-zero gradient to a step of +-lr: rounding-level
    differences of near-zero gradients become 2 lr differences of single weights): the default mode within max(1.2e-2, three
    times the drift of bf16x6 in the same run).  Step 3 is printed, not
    asserted - the trajectories separate at the same rate in both split modes (measured: f16x3 5e-2, bf16x6 4e-1 on the
    most sensitive term), i.e. it is the optimiser's sensitivity, not the arithmetic of a mode."""
    import test_gpu_model as M
    from modules.discriminator import GCNDiscriminatorDecouple
    from modules.model import Counter3DDisc, Counter3DModel
    from xas_amd.engine import TrainStep
    from xas_amd.optim import FusedAdam
    import inputs as gi
    cfg = gi.model_params('S2', cam_ids=(0, 1))
    full = {'model_params': cfg, 'train_params': {'lr_kp_detector': 1e-4, 'lr_discriminator': 1e-4}}
    xg = {k: torch.from_numpy(v).cuda() for k, v in gi.synthetic_batch(2, [0, 1], seed=95).items()}
    runs = {}
    for mode in ('f32', 'f16x3', 'bf16x6'):
        with precision_mode(mode):
            reg, phys, _, _ = M._hip_models('S2', (0, 1))
            disc = gi.seeded_fill_(GCNDiscriminatorDecouple(cfg['smpl_disc_params']), seed=9).cuda().train()
            disc.header.p = 0.0
            gen, dis = Counter3DModel(cfg, reg, None, None, phys), Counter3DDisc(cfg, disc, None, None)
            opt_det = FusedAdam(list(reg.parameters()) + list(phys.parameters()), lr=1e-4, betas=(0.5, 0.999))
            opt_disc = FusedAdam(disc.parameters(), lr=1e-4, betas=(0.5, 0.999))
            step = TrainStep(full, gen, dis, opt_det, opt_disc)
            hist = []
            for _ in range(3):
                ld, lk, tot, _ = step(xg)
                hist.append([float(ld)] + [float(lk[k].mean()) for k in sorted(lk)])
            torch.cuda.synchronize()
            runs[mode] = torch.tensor(hist, dtype=torch.float64)
    errs = {}
    for mode in ('f16x3', 'bf16x6'):
        rel_e = (runs[mode] - runs['f32']).abs() / (runs['f32'].abs() + 1e-6)
        errs[mode] = float(rel_e.max())
        print(mode, 'vs f32 over 3 free-running steps: relative loss differences per step\n', rel_e)
        # (step 2 is a SENSITIVITY, not an accuracy: one Adam update turns rounding-level differences of near-zero gradients
        # into 2 lr steps of single weights; measured 2e-4 .. 6e-3 depending only on which launches share a pass - r04's joint
        # prefix pass moved it from 3e-3 to 5.8e-3 with the arithmetic of every product unchanged - hence a bar of 3e-2)
        errs[mode + '_steps'] = rel_e
        # (r05: the split modes take the head's first pass from the final convolution's epilogue, the exact-fp32 kernels run the
        # head's own two passes - another summation order of the soft-argmax sums: 4.6e-6 / 5.3e-6 measured, was 3e-6)
        assert float(rel_e[0].max()) < 1e-5, (mode, rel_e)
    # step 2 against a YARDSTICK from the same state instead of a loose constant: the default mode may not drift
    # further from exact fp32 than 1.2e-2 (twice the largest value measured for it, 5.8e-3) or three times what the range-free
    # six-product mode drifts in the same run, whichever is larger
    s2 = {m: float(errs[m + '_steps'][1].max()) for m in ('f16x3', 'bf16x6')}
    print('step-2 drift vs exact fp32:', s2)
    assert s2['bf16x6'] < 3e-2, s2
    assert s2['f16x3'] < max(1.2e-2, 3.0 * s2['bf16x6']), s2


def test_f16x3_weight_range_is_checked():
    """Weights are split as 2^10 w: a weight of magnitude 64 or more cannot be represented.  The preparation kernels raise a
    device flag (xas_f16_weight_overflow; engine.TrainStep polls it) instead of producing inf / NaN silently."""
    from xas_amd import layers as L
    from xas_amd._lib import query
    query('xas_f16_weight_overflow', 1)
    m = L.Conv2d(64, 64, 3, 1, 1, bias=False).cuda()
    x = torch.randn(2, 64, 16, 16).cuda()
    with precision_mode('f16x3'), torch.no_grad():
        m(x)
        torch.cuda.synchronize()
        assert query('xas_f16_weight_overflow', 1) == 0
        m.weight[3, 5, 1, 1] = 100.0
        from xas_amd import ops_nn
        ops_nn.bump_weights_epoch()
        m(x)
        torch.cuda.synchronize()
        assert query('xas_f16_weight_overflow', 1) == 1
        assert query('xas_f16_weight_overflow', 0) == 0          # cleared by the previous call


def test_train_step_reports_a_weight_leaving_the_range_within_a_few_steps():
    """engine.TrainStep reads the flag WITHOUT a synchronisation, one step late (xas_f16_weight_overflow_peek + an asynchronous
    copy to pinned memory): a weight that an update pushes beyond |w| < 64 stops the training a step or two later, not at the
    next 64-step poll."""
    from xas_amd import engine
    from xas_amd._lib import query
    from xas_amd.synthetic import model_config, synthetic_batch
    query('xas_f16_weight_overflow', 1)
    cfg = model_config('HM36_Multi_SurS2')
    cfg['model_params']['cam_id_list'] = [0]
    torch.manual_seed(3)
    model, disc, od, odisc = engine.prepare_model(cfg)
    model.cuda().train(), disc.cuda().train()
    step = engine.TrainStep(cfg, model, disc, od, odisc)
    x = synthetic_batch(2, [0], torch.device('cuda'), seed=4)
    with precision_mode('f16x3'):
        for _ in range(3):
            step(x)                                                   # in range: nothing is raised
        torch.cuda.synchronize()
        with torch.no_grad():
            model.regressor.net.backbone.layer2[1].conv2.weight[7, 9, 1, 1] = 300.0      # "an update" that leaves the range
        od._epoch[0] += 1
        with pytest.raises(RuntimeError, match='left the range'):
            for _ in range(6):                                        # far fewer than the 64 steps of the synchronising poll
                step(x)
                torch.cuda.synchronize()
        assert query('xas_f16_weight_overflow', 1) == 1               # (the peek does not clear the flag)
