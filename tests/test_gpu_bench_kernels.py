"""Full-size value checks of the kernels bench.py actually runs.

The layer sweep of test_gpu_parity_r3.py runs at two images, where the dispatch picks the small-problem tiles; the B = 32
step of bench.py runs 384- / 256-image passes on 64 x 256 tiles, tap re-use kernels and the split-K weight-gradient
kernels.  Here the heaviest launches of `profiles/r04_conv_shapes_b32.txt` run through the C ABI AT THE BENCH'S N, with
the operand maxima the hot path would pass (the f16x3 default), and are checked against float64 evaluated from the
operands: forward / data gradient on 4 096 randomly sampled output elements, the weight gradient on a randomly sampled
32 x 32 block of (output channel, input channel) pairs for every filter tap (each element a sum over all N * Ho * Wo rows).
The float64 side is plain torch indexing / einsum of the definition - no oracle module, no HIP kernel of this repository.
reference: modules/integral_base_modules/deconv_head.py:24-35 (deconvolutions + final 1x1), resnet.py:16-47,
modules/physique_network.py:41-50.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

S = 4096            # sampled output elements per pass
BLK = 32            # weight gradient: BLK x BLK (cout, cin) pairs, every tap


def _gen(seed):
    return torch.Generator(device='cuda').manual_seed(seed)


def _operands(n, hi, wi, cin, cout, k, stride, pad, seed, need_dy=True):
    """x [n,cin,hi,wi], w [cout,cin,k,k] (OIHW), dy [n,cout,ho,wo]: channels_last device tensors with channel offsets and
    a per-channel spread (batch-norm outputs look like that, not like N(0,1))."""
    g = _gen(seed)
    ho, wo = (hi + 2 * pad - k) // stride + 1, (wi + 2 * pad - k) // stride + 1
    cl = torch.channels_last
    x = torch.randn(n, cin, hi, wi, device='cuda', generator=g).contiguous(memory_format=cl)
    x.mul_(0.5 + torch.rand(1, cin, 1, 1, device='cuda', generator=g)).add_(0.3)
    w = torch.randn(cout, cin, k, k, device='cuda', generator=g) / (cin * k * k) ** 0.5
    dy = None
    if need_dy:
        dy = torch.randn(n, cout, ho, wo, device='cuda', generator=g).contiguous(memory_format=cl)
        dy.mul_(1e-3 * (0.2 + torch.rand(1, cout, 1, 1, device='cuda', generator=g)))      # gradient magnitudes, heavy spread
    return x, w, dy, ho, wo


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def _fwd_samples(x, w, bias, stride, pad, ho_, wo_, seed):
    """-> (index tuple into y [n,cout,ho,wo], float64 values of the convolution there)."""
    g = _gen(seed)
    N, Ci, H, W = x.shape
    Co, _, R, Sx = w.shape
    ri = lambda hi: torch.randint(0, hi, (S,), device='cuda', generator=g)
    n, co, ho, wo = ri(N), ri(Co), ri(ho_), ri(wo_)
    acc = torch.zeros(S, device='cuda', dtype=torch.float64)
    for r in range(R):
        for s in range(Sx):
            hi, wi = ho * stride - pad + r, wo * stride - pad + s
            ok = (hi >= 0) & (hi < H) & (wi >= 0) & (wi < W)
            xv = x[n, :, hi.clamp(0, H - 1), wi.clamp(0, W - 1)].double()            # [S, Ci]
            acc += (xv * w[co, :, r, s].double()).sum(1) * ok
    if bias is not None:
        acc += bias[co].double()
    return (n, co, ho, wo), acc


def _dgrad_samples(dy, w, stride, pad, hi_, wi_, seed):
    """-> (index tuple into dx [n,cin,hi,wi], float64 values of the data gradient there)."""
    g = _gen(seed)
    N, Co, Ho, Wo = dy.shape
    _, Ci, R, Sx = w.shape
    ri = lambda hi: torch.randint(0, hi, (S,), device='cuda', generator=g)
    n, ci, hi, wi = ri(N), ri(Ci), ri(hi_), ri(wi_)
    acc = torch.zeros(S, device='cuda', dtype=torch.float64)
    for r in range(R):
        for s in range(Sx):
            th, tw = hi + pad - r, wi + pad - s
            ho, wo = torch.div(th, stride, rounding_mode='floor'), torch.div(tw, stride, rounding_mode='floor')
            ok = (th % stride == 0) & (tw % stride == 0) & (ho >= 0) & (ho < Ho) & (wo >= 0) & (wo < Wo)
            gv = dy[n, :, ho.clamp(0, Ho - 1), wo.clamp(0, Wo - 1)].double()          # [S, Co]
            acc += (gv * w[:, ci, r, s].t().double()).sum(1) * ok
    return (n, ci, hi, wi), acc


def _wgrad_block(x, dy, k, stride, pad, seed):
    """-> (cout index [BLK], cin index [BLK], float64 dw[cout_idx][:, cin_idx] for every tap: [BLK, BLK, k, k])."""
    g = _gen(seed)
    N, Ci, H, W = x.shape
    _, Co, Ho, Wo = dy.shape
    co = torch.randperm(Co, device='cuda', generator=g)[:min(BLK, Co)].sort().values
    ci = torch.randperm(Ci, device='cuda', generator=g)[:min(BLK, Ci)].sort().values
    dys = dy[:, co].double()                                                         # [N, b, Ho, Wo]
    xs = torch.nn.functional.pad(x[:, ci].double(), (pad, pad, pad, pad))            # [N, b, H + 2p, W + 2p]
    out = torch.empty(len(co), len(ci), k, k, device='cuda', dtype=torch.float64)
    for r in range(k):
        for s in range(k):
            win = xs[:, :, r:r + stride * (Ho - 1) + 1:stride, s:s + stride * (Wo - 1) + 1:stride]
            out[:, :, r, s] = torch.einsum('nahw,nbhw->ab', dys, win)
    return co, ci, out


def _shape(F, n, hi, wi, cin, cout, k, stride, pad, ho, wo):
    return F._shape(n, hi, wi, cin, cout, k, k, stride, pad, ho, wo)


# (what, n, hi, wi, cin, cout, k, stride, pad): the heaviest conv launches of the B = 32 HM36 step
# (profiles/r04_conv_shapes_b32.txt: ms per step in the comment), at the image count of the pass that runs them
FWD_CASES = [
    ('final 1x1 + bias, 3 x 4 cameras x 32 (3.4 ms)', 384, 64, 64, 256, 1152, 1, 1, 0),
    ('layer1 conv3 P -> 4P (short K, 2.2 ms / 4)', 384, 64, 64, 64, 256, 1, 1, 0),
    ('layer1 conv2 3x3 tap kernel', 384, 64, 64, 64, 64, 3, 1, 1),
    ('layer3 conv2 3x3 tap kernel, 16 x 16 maps', 384, 16, 16, 256, 256, 3, 1, 1),
    ('deconv data gradient = forward conv 4x4 s2 (2.1 ms)', 256, 64, 64, 256, 256, 4, 2, 1),
    ('physique 3x3 64 -> 32 on 256 x 256', 128, 256, 256, 64, 32, 3, 1, 1),
]


@pytest.mark.parametrize('what,n,hi,wi,cin,cout,k,stride,pad', FWD_CASES, ids=[c[0].split(' (')[0].replace(' ', '_') for c in FWD_CASES])
def test_forward_at_bench_size_vs_float64_samples(what, n, hi, wi, cin, cout, k, stride, pad):
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr
    x, w, _, ho, wo = _operands(n, hi, wi, cin, cout, k, stride, pad, seed=cin + cout + k + n, need_dy=False)
    bias = torch.randn(cout, device='cuda', generator=_gen(5)) if k == 1 and cout == 1152 else None
    shp = F.shape_with_maxima(_shape(F, n, hi, wi, cin, cout, k, stride, pad, ho, wo), x)
    y = torch.full((n, cout, ho, wo), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    cache = F._PackCache()
    call('xas_conv_fwd', ptr(x), ptr(cache.get(w, 0, shp)), ptr(bias), ptr(y), shp)
    idx, ref = _fwd_samples(x, w, bias, stride, pad, ho, wo, seed=11)
    e = _rel(y[idx], ref)
    print('%s: forward vs float64 on %d samples: %.2e' % (what, S, e))
    assert bool(torch.isfinite(y).all())
    assert e < 3e-6, (what, e)


# bottleneck conv + norm statistics in the epilogue, G = 3 x 4 cameras
@pytest.mark.parametrize('n,hi,cin,cout,k,G', [(384, 64, 64, 256, 1, 12), (384, 64, 256, 64, 1, 12), (384, 64, 64, 64, 3, 12)])
def test_forward_with_norm_statistics_at_bench_size(n, hi, cin, cout, k, G):
    """xas_conv_fwd_bnstats at the bench's size: sampled outputs against float64 from the operands, per-group mean / biased
    variance against float64 statistics of the written output."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr, query
    pad = k // 2
    x, w, _, ho, wo = _operands(n, hi, hi, cin, cout, k, 1, pad, seed=3 + cin + cout + k, need_dy=False)
    shp = F.shape_with_maxima(_shape(F, n, hi, hi, cin, cout, k, 1, pad, ho, wo), x)
    y = torch.full((n, cout, ho, wo), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
    ws = torch.empty(query('xas_conv_fwd_bnstats_workspace_floats', shp, G), device='cuda')
    mean = torch.empty(G, cout, device='cuda'); var = torch.empty(G, cout, device='cuda')
    rm, rv = torch.zeros(cout, device='cuda'), torch.ones(cout, device='cuda')
    cache = F._PackCache()
    call('xas_conv_fwd_bnstats', ptr(x), ptr(cache.get(w, 0, shp)), ptr(y), shp, G, None, ptr(mean), ptr(var), cout, None,
         ptr(ws), ptr(rm), ptr(rv), 0.1)
    idx, ref = _fwd_samples(x, w, None, 1, pad, ho, wo, seed=12)
    e = _rel(y[idx], ref)
    rows = y.permute(0, 2, 3, 1).reshape(G, -1, cout)
    mean64 = torch.stack([rows[g].double().mean(0) for g in range(G)])
    var64 = torch.stack([rows[g].double().var(0, unbiased=False) for g in range(G)])
    em = float((mean.double() - mean64).abs().max() / (mean64.abs().max() + var64.max().sqrt()))
    ev = float(((var.double() - var64).abs() / var64).max())
    print('fwd_bnstats %s: samples %.2e, mean %.2e, var %.2e' % ((n, hi, cin, cout, k), e, em, ev))
    assert e < 3e-6 and em < 2e-6 and ev < 2e-5, (e, em, ev)


DGRAD_CASES = [
    ('final 1x1 data gradient (1.9 ms)', 256, 64, 64, 256, 1152, 1, 1, 0),
    ('deconv forward = data gradient 4x4 s2, 384 images (2.1 ms)', 384, 64, 64, 256, 256, 4, 2, 1),
    ('layer1 conv2 3x3 tap kernel', 256, 64, 64, 64, 64, 3, 1, 1),
    ('layer4 conv3 1x1 512 -> 2048', 256, 8, 8, 512, 2048, 1, 1, 0),
]


@pytest.mark.parametrize('what,n,hi,wi,cin,cout,k,stride,pad', DGRAD_CASES, ids=[c[0].split(' (')[0].replace(' ', '_') for c in DGRAD_CASES])
def test_data_gradient_at_bench_size_vs_float64_samples(what, n, hi, wi, cin, cout, k, stride, pad):
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr
    x, w, dy, ho, wo = _operands(n, hi, wi, cin, cout, k, stride, pad, seed=1 + cin + cout + k + n)
    shp = F.shape_with_maxima(_shape(F, n, hi, wi, cin, cout, k, stride, pad, ho, wo), dy)
    dx = torch.full_like(x, float('nan'))
    cache = F._PackCache()
    call('xas_conv_dgrad', ptr(dy), ptr(cache.get(w, 1, shp)), ptr(dx), shp)
    idx, ref = _dgrad_samples(dy, w, stride, pad, hi, wi, seed=13)
    e = _rel(dx[idx], ref)
    print('%s: data gradient vs float64 on %d samples: %.2e' % (what, S, e))
    assert bool(torch.isfinite(dx).all())
    assert e < 3e-6, (what, e)


def test_masked_accumulating_data_gradient_at_bench_size():
    """xas_conv_dgrad_acc_masked (256, 64, 64, 256 <- 64): dx = dgrad(dy, W) + relu'(mask) * dprev - the block-input gradient
    of every bottleneck without projection (profiles/r04_conv_shapes_b32.txt: 68 TFLOP/s, the short-K data gradient)."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr
    n, hi, cin, cout = 256, 64, 256, 64
    x, w, dy, ho, wo = _operands(n, hi, hi, cin, cout, 1, 1, 0, seed=77)
    g = _gen(78)
    dprev = (torch.randn(n, cin, hi, hi, device='cuda', generator=g) * 1e-3).contiguous(memory_format=torch.channels_last)
    active = torch.rand(n, hi, hi, cin, device='cuda', generator=g) > 0.4             # NHWC order, like the mask bytes
    a4 = active.reshape(-1, 4).to(torch.uint8)
    mask = (a4[:, 0] | (a4[:, 1] << 1) | (a4[:, 2] << 2) | (a4[:, 3] << 3)).contiguous()
    shp = F.shape_with_maxima(_shape(F, n, hi, hi, cin, cout, 1, 1, 0, ho, wo), dy)
    out = torch.full_like(x, float('nan'))
    cache = F._PackCache()
    call('xas_conv_dgrad_acc_masked', ptr(dy), ptr(cache.get(w, 1, shp)), ptr(out), shp, ptr(dprev), ptr(mask))
    idx, ref = _dgrad_samples(dy, w, 1, 0, hi, hi, seed=14)
    nn_, ci, h_, w_ = idx
    ref = ref + dprev[idx].double() * active[nn_, h_, w_, ci]
    e = _rel(out[idx], ref)
    print('masked accumulating data gradient vs float64: %.2e' % e)
    assert e < 3e-6, e


WGRAD_CASES = [
    ('final 1x1 (4.2 ms, the largest conv launch of the step)', 256, 64, 64, 256, 1152, 1, 1, 0),
    ('deconv 4x4 s2 (3.2 ms)', 256, 64, 64, 256, 256, 4, 2, 1),
    ('physique 3x3 64 -> 32 on 256 x 256 (3.6 ms)', 128, 256, 256, 64, 32, 3, 1, 1),
    ('layer3 conv2 3x3 on 16 x 16 maps (2.8 ms / 5)', 256, 16, 16, 256, 256, 3, 1, 1),
    ('layer1 conv2 3x3 tap kernel (1.9 ms / 3)', 256, 64, 64, 64, 64, 3, 1, 1),
    ('layer1 conv3 64 -> 256 (1.7 ms / 4)', 256, 64, 64, 64, 256, 1, 1, 0),
    ('layer4 conv1 2048 -> 512 on 8 x 8 maps', 256, 8, 8, 2048, 512, 1, 1, 0),
]


@pytest.mark.parametrize('what,n,hi,wi,cin,cout,k,stride,pad', WGRAD_CASES, ids=[c[0].split(' (')[0].replace(' ', '_') for c in WGRAD_CASES])
def test_weight_gradient_at_bench_size_vs_float64_block(what, n, hi, wi, cin, cout, k, stride, pad):
    """xas_conv_wgrad_acc (the form the side stream uses: grad arena += dw) with both operand maxima: a 32 x 32 block of
    (cout, cin) pairs, every tap, against float64 sums over all N * Ho * Wo rows."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr, query
    x, w, dy, ho, wo = _operands(n, hi, wi, cin, cout, k, stride, pad, seed=2 + cin + cout + k + n)
    prev = torch.randn(cout, cin, k, k, device='cuda', generator=_gen(9)) * 1e-2
    buf = prev.clone()
    shp = F.shape_with_maxima(_shape(F, n, hi, wi, cin, cout, k, stride, pad, ho, wo), dy, x)
    ws = torch.empty(max(1, query('xas_conv_wgrad_workspace_floats', shp)), device='cuda')
    call('xas_conv_wgrad_acc', ptr(x), ptr(dy), ptr(buf), ptr(ws), shp)
    co, ci, ref = _wgrad_block(x, dy, k, stride, pad, seed=15)
    got = (buf - prev)[co][:, ci]
    e = _rel(got, ref)
    print('%s: weight gradient vs float64 on a %d x %d x %d x %d block: %.2e' % (what, len(co), len(ci), k, k, e))
    assert bool(torch.isfinite(buf).all())
    # (sums of 0.26 - 8.4 million products: the bar of the isolated tests at these sizes, tests/test_gpu_kernels_isolated.py)
    assert e < 1e-5, (what, e)
