"""igemm_x6p_kernel (csrc/conv_x6.hip): the persistent 64 x 256-tile kernel that runs the 1x1 stride-1 convolutions with wide
outputs in the f16x3 arithmetic - a block walks over several M-tiles and its loads run ahead into the next tile.

Checked two ways through the C ABI: (1) against float64 evaluated from the operands on sampled outputs (the bar of every conv
kernel, 3e-6); (2) BIT-IDENTICAL to igemm_x6_kernel<64, 256> (tuning bit 26 switches the persistent kernel off): same
fragments, same MFMA order, same epilogue - for tile counts that are / are not multiples of the tiles per block, one and several
N-tiles, the norm-statistics epilogue and the accumulate-and-mask epilogue.
reference: modules/integral_base_modules/resnet.py:16-47 (torchvision Bottleneck conv1 / conv3), deconv_head.py:34-35."""
import pytest
import torch

from test_gpu_bench_kernels import _dgrad_samples, _fwd_samples, _gen, _operands, _rel, _shape

pytestmark = pytest.mark.gpu

NO_PERSIST = 1 << 26
ANY_K = 1 << 27          # the dispatch keeps K > 128 on the one-tile kernel (no gain there): the tests run those shapes through
                        # the persistent kernel as well


def _with_tuning(bits, fn):
    from xas_amd._lib import query
    query('xas_set_tuning', bits)
    try:
        return fn()
    finally:
        query('xas_set_tuning', 0)


# (n, h, w, cin, cout): M = n h w rows; 64-row tiles; the launcher takes tiles / 3072 (1 ... 8) tiles per block
FWD = [(384, 64, 64, 64, 256),        # layer1 conv3 at the bench's size: 24 576 tiles, 8 per block
       (96, 32, 32, 128, 512),        # 1 536 x 2 tiles: 1 per block
       (37, 40, 40, 64, 256),         # 925 tiles (odd count)
       (101, 64, 64, 64, 256),        # 6 464 tiles -> 2 per block, an even count
       (203, 40, 40, 64, 256),        # 5 075 tiles -> 1 per block... and
       (317, 40, 40, 128, 256),       # 7 925 tiles -> 2 per block, ODD count: the last block has one tile
       (45, 36, 36, 64, 256),         # 58 320 rows = 911.25 tiles: a ragged last tile (register epilogue, out-of-range rows not loaded)
       (333, 36, 36, 128, 256),       # 431 568 rows = 6 743.25 tiles -> 2 per block, ragged
       (64, 64, 64, 256, 1024)]       # four N-tiles, 4 096 x 4 tiles -> 5 per block: the last group of an N-tile has one tile


@pytest.mark.parametrize('n,h,w,cin,cout', FWD)
def test_persistent_forward_vs_float64_and_bit_identical_to_one_tile_kernel(n, h, w, cin, cout):
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr
    x, wt, _, ho, wo = _operands(n, h, w, cin, cout, 1, 1, 0, seed=n + cin + cout, need_dy=False)
    bias = torch.randn(cout, device='cuda', generator=_gen(9)) if cout == 1024 else None
    shp = F.shape_with_maxima(_shape(F, n, h, w, cin, cout, 1, 1, 0, ho, wo), x)
    cache = F._PackCache()
    wp = cache.get(wt, 0, shp)

    def run():
        y = torch.full((n, cout, ho, wo), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
        call('xas_conv_fwd', ptr(x), ptr(wp), ptr(bias), ptr(y), shp)
        return y
    y = _with_tuning(ANY_K, run)
    y1 = _with_tuning(NO_PERSIST, run)
    idx, ref = _fwd_samples(x, wt, bias, 1, 0, ho, wo, seed=21)
    e = _rel(y[idx], ref)
    print('persistent forward %s: %.2e vs float64' % ((n, h, w, cin, cout), e))
    assert bool(torch.isfinite(y).all()) and e < 3e-6, e
    assert torch.equal(y, y1)


@pytest.mark.parametrize('n,hi,cin,cout,G', [(384, 64, 64, 256, 12), (36, 16, 256, 1024, 12), (60, 8, 512, 2048, 12)])
def test_persistent_forward_with_norm_statistics(n, hi, cin, cout, G):
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr, query
    x, wt, _, ho, wo = _operands(n, hi, hi, cin, cout, 1, 1, 0, seed=5 + cin + cout, need_dy=False)
    shp = F.shape_with_maxima(_shape(F, n, hi, hi, cin, cout, 1, 1, 0, ho, wo), x)
    cache = F._PackCache()
    wp = cache.get(wt, 0, shp)

    def run():
        y = torch.full((n, cout, ho, wo), float('nan'), device='cuda').contiguous(memory_format=torch.channels_last)
        ws = torch.empty(query('xas_conv_fwd_bnstats_workspace_floats', shp, G), device='cuda')
        mean = torch.empty(G, cout, device='cuda'); var = torch.empty(G, cout, device='cuda')
        rm, rv = torch.zeros(cout, device='cuda'), torch.ones(cout, device='cuda')
        call('xas_conv_fwd_bnstats', ptr(x), ptr(wp), ptr(y), shp, G, None, ptr(mean), ptr(var), cout, None, ptr(ws), ptr(rm), ptr(rv), 0.1)
        return y, mean, var, rm, rv
    a = _with_tuning(ANY_K, run)
    b = _with_tuning(NO_PERSIST, run)
    y, mean, var = a[:3]
    idx, ref = _fwd_samples(x, wt, None, 1, 0, ho, wo, seed=22)
    rows = y.permute(0, 2, 3, 1).reshape(G, -1, cout)
    mean64 = torch.stack([rows[g].double().mean(0) for g in range(G)])
    var64 = torch.stack([rows[g].double().var(0, unbiased=False) for g in range(G)])
    e = _rel(y[idx], ref)
    em = float((mean.double() - mean64).abs().max() / (mean64.abs().max() + var64.max().sqrt()))
    ev = float(((var.double() - var64).abs() / var64).max())
    assert e < 3e-6 and em < 2e-6 and ev < 2e-5, (e, em, ev)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize('n,hi,cin,cout', [(256, 64, 256, 64), (77, 32, 512, 128), (50, 16, 1024, 256)])
def test_persistent_masked_accumulating_data_gradient(n, hi, cin, cout):
    """dx = dgrad(dy, W) + relu'(mask) * dprev through the persistent kernel (K = cout of the layer: 64 ... 256)."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr
    x, wt, dy, ho, wo = _operands(n, hi, hi, cin, cout, 1, 1, 0, seed=70 + cin)
    g = _gen(79)
    dprev = (torch.randn(n, cin, hi, hi, device='cuda', generator=g) * 1e-3).contiguous(memory_format=torch.channels_last)
    active = torch.rand(n, hi, hi, cin, device='cuda', generator=g) > 0.4
    a4 = active.reshape(-1, 4).to(torch.uint8)
    mask = (a4[:, 0] | (a4[:, 1] << 1) | (a4[:, 2] << 2) | (a4[:, 3] << 3)).contiguous()
    shp = F.shape_with_maxima(_shape(F, n, hi, hi, cin, cout, 1, 1, 0, ho, wo), dy)
    cache = F._PackCache()
    wp = cache.get(wt, 1, shp)

    def run():
        out = torch.full_like(x, float('nan'))
        call('xas_conv_dgrad_acc_masked', ptr(dy), ptr(wp), ptr(out), shp, ptr(dprev), ptr(mask))
        return out
    out = _with_tuning(ANY_K, run)
    out1 = _with_tuning(NO_PERSIST, run)
    idx, ref = _dgrad_samples(dy, wt, 1, 0, hi, hi, seed=15)
    nn_, ci, h_, w_ = idx
    ref = ref + dprev[idx].double() * active[nn_, h_, w_, ci]
    e = _rel(out[idx], ref)
    assert bool(torch.isfinite(out).all()) and e < 3e-6, e
    assert torch.equal(out, out1)


def test_persistent_plain_data_gradient_long_k():
    """layer3 conv1 data gradient (K = 256 -> 1024 columns): the register epilogue inside the tile loop."""
    from xas_amd import ops_nn as F
    from xas_amd._lib import call, ptr
    n, hi, cin, cout = 130, 16, 1024, 256
    x, wt, dy, ho, wo = _operands(n, hi, hi, cin, cout, 1, 1, 0, seed=31)
    shp = F.shape_with_maxima(_shape(F, n, hi, hi, cin, cout, 1, 1, 0, ho, wo), dy)
    cache = F._PackCache()
    wp = cache.get(wt, 1, shp)

    def run():
        dx = torch.full_like(x, float('nan'))
        call('xas_conv_dgrad', ptr(dy), ptr(wp), ptr(dx), shp)
        return dx
    dx = _with_tuning(ANY_K, run)
    dx1 = _with_tuning(NO_PERSIST, run)
    idx, ref = _dgrad_samples(dy, wt, 1, 0, hi, hi, seed=16)
    e = _rel(dx[idx], ref)
    assert bool(torch.isfinite(dx).all()) and e < 3e-6, e
    assert torch.equal(dx, dx1)
