"""Round-3 parity tightening (VERDICT r02 "tighten parity where it is loose"):

* every one of the detector's 170 parameter gradients against the imported reference (golden `detector_allgrads`), with
  a per-tensor tolerance derived from the reference's own fp32-vs-fp64 distance instead of 3e-2 on eight slices;
* every convolution / transposed-convolution SHAPE of the detector and of the physique net (real channel counts,
  spatial sizes, strides - N = 2) through all three passes against a float64 convolution at 3e-6: a wrong tap or a
  wrong tile edge in any of the layer shapes fails here;
* the HM36 four-camera wiring golden: the joined G = 8 real + pseudo detector pass against the reference's eight calls;
* one full optimisation step at the BASELINE size (B = 32 x 4 cameras): camera-batched == one call per camera.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

import inputs as gi
from conftest import check_all_grads, golden

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_detector_all_parameter_gradients():
    from modules.keypoint_detector_integral import KPDetector3D
    from oracle import step as ostep
    g = golden('detector_allgrads')
    ora = gi.seeded_fill_(ostep.Regressor('resnet', 18, 64), seed=61)
    hip = KPDetector3D('resnet', 18, 64)
    hip.load_state_dict(ora.state_dict(), strict=True)
    hip.cuda().train()
    x = T(gi.synthetic_batch(2, [0], seed=63)['cam_0_img']).cuda()
    kps, _ = hip(x)
    assert float((kps.cpu() - T(g['kps'])).abs().max()) < 1e-4               # north-star bar
    assert float((kps.cpu() - T(g['kps_f64'])).abs().max()) < 1e-4           # and against the float64 evaluation
    gw = T(np.random.Generator(np.random.PCG64(64)).standard_normal(kps.shape).astype(np.float32)).cuda()
    (kps * gw).sum().backward()
    torch.cuda.synchronize()
    # two fp32 evaluations of this graph each sit `dev` from the float64 one (ReLU / max-pool decisions on near-ties):
    # their mutual distance is bounded by the sum -> factor 4 with a 3e-3 floor for the well-conditioned tensors
    worst = check_all_grads([(n, p.grad) for n, p in hip.named_parameters()], g, 3e-3, 4.0, 'hip')
    print('worst error / tolerance over 170 tensors: %.2f' % worst)


def _layer_shapes():
    """(kind, cin, cout, k, stride, pad, h, w) of every conv of the detector and the physique net, de-duplicated; taken
    from the oracle modules with forward hooks on a 1 x 256 x 256 input (geometry only, CPU, meta-cheap)."""
    from oracle.nets import Detector, PhysiqueNet
    seen = []

    def hook(m, inp, out):
        x = inp[0]
        if isinstance(m, torch.nn.ConvTranspose2d):
            key = ('deconv', m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], m.padding[0], x.shape[2], x.shape[3])
        else:
            key = ('conv', m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], m.padding[0], x.shape[2], x.shape[3])
        if key not in seen:
            seen.append(key)

    for net, inp in ((Detector(18, 64), torch.zeros(1, 3, 256, 256)), (PhysiqueNet([32, 64, 128]), torch.zeros(1, 1, 256, 256))):
        hs = [m.register_forward_hook(hook) for m in net.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))]
        net.eval()
        with torch.no_grad():
            net(inp)
        for h in hs:
            h.remove()
    return seen


def test_every_layer_shape_all_passes_vs_float64():
    from xas_amd import layers as L
    shapes = _layer_shapes()
    assert len(shapes) >= 30
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    worst = {}
    for kind, cin, cout, k, stride, pad, h, w in shapes:
        g = torch.Generator().manual_seed(cin * 7 + cout * 3 + k + h)
        n = 2
        x = torch.randn(n, cin, h, w, generator=g)
        if kind == 'conv':
            wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
            m = L.Conv2d(cin, cout, k, stride, pad, bias=False).cuda()
            xc, wc = x.double().requires_grad_(True), wt.double().requires_grad_(True)
            yc = TF.conv2d(xc, wc, None, stride, pad)
        else:
            wt = torch.randn(cin, cout, k, k, generator=g) / (cin * k * k / stride ** 2) ** 0.5
            m = L.ConvTranspose2d(cin, cout, k, stride, pad).cuda()
            xc, wc = x.double().requires_grad_(True), wt.double().requires_grad_(True)
            yc = TF.conv_transpose2d(xc, wc, None, stride, pad)
        gy = torch.randn(yc.shape, generator=g)
        (yc * gy.double()).sum().backward()
        with torch.no_grad():
            m.weight.copy_(wt)
        xg = x.cuda().requires_grad_(True)
        yg = m(xg)
        (yg * gy.cuda()).sum().backward()
        torch.cuda.synchronize()
        e = (rel(yg, yc), rel(xg.grad, xc.grad), rel(m.weight.grad, wc.grad))
        key = (kind, cin, cout, k, stride, pad, h, w)
        worst[key] = e
        assert e[0] < 3e-6 and e[1] < 3e-6 and e[2] < 3e-6, (key, e)
    print('layer shapes checked: %d, worst fwd/dgrad/wgrad %.1e %.1e %.1e' % (
        len(worst), max(v[0] for v in worst.values()), max(v[1] for v in worst.values()), max(v[2] for v in worst.values())))


def test_model_wiring_hm36_four_cameras():
    from test_gpu_model import _check_wiring, _yaml_params
    cams = (0, 1, 2, 3)
    mp = _yaml_params('HM36_Multi_SurS1', cams)
    mp['loss_config']['recons_loss']['weight'] = 0.02
    mp['loss_config']['physique_recons_loss']['weight'] = 0.02
    g, reg, phys = _check_wiring('model_HM36_Multi_SurS1_4cam', mp, cams, 87)
    assert float(g['loss_reconstruction']) > 1e-4
    assert rel(phys.encoder[0][0].weight.grad, T(g['g_phys_enc0_w'])) < 3e-2
    assert rel(dict(reg.named_parameters())['net.backbone.layer1.0.conv2.weight'].grad[:8], T(g['g_l1c2'])) < 5e-2


def _schedule_run(name, batch, batched, mode, monkeypatch):
    """One step of BASELINE config `name` from the seeded initial state (planted depth peaks, _stepcheck.build_step) in one
    schedule / precision -> captured result (gradient arenas as Adam consumes them, losses, peak indices, parameters)."""
    import modules.model as mm
    import _stepcheck as sc
    from conftest import precision_mode
    monkeypatch.setattr(mm, 'CAM_BATCH', batched)
    monkeypatch.setattr(mm, 'JOIN_PSEUDO', batched)
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    step, x = sc.build_step(name, batch)
    with precision_mode(mode):
        r = sc.run_captured(step, x)
    r['peak_gb'] = torch.cuda.max_memory_allocated() / 2**30
    r['losses'] = dict(zip(r['loss_names'], (float(v) for v in r['loss'])))
    assert bool(torch.isfinite(r['loss']).all()) and bool(torch.isfinite(r['det']).all()) and bool(torch.isfinite(r['params']).all())
    del step, x
    torch.cuda.empty_cache()
    return r


def _schedules_agree(a, b, tol, grad_bar, what):
    """Two schedules / arithmetics of the SAME step: every loss term within `tol` relative (the adversarial term, which sees
    the updated discriminator, 10 x), the int64 depth-peak indices identical, and the
    detector gradient arena within `grad_bar` in norm.  The bar comes from the measured distance of two exact-fp32 evaluations
    of this gradient with different summation orders (6.5e-3 .. 9e-3, bench.py variant_check) - NOT a sign-flip count after
    Adam, which measures how many |g| sit near zero, not how far the gradients are apart."""
    for k, v in b['losses'].items():
        t = 10 * tol if k == 'smpl_gen' else tol
        assert abs(a['losses'][k] - v) <= t * abs(v) + 1e-7, (what, k, a['losses'][k], v)
    # (both schedules run the reference's detector calls in the reference's order: disc real, gen real, gen pseudo)
    assert a['peaks'].shape == b['peaks'].shape and a['peaks'].numel() > 0
    n_diff = int((a['peaks'] != b['peaks']).sum())
    assert n_diff == 0, '%s: %d of %d depth-peak indices differ' % (what, n_diff, a['peaks'].numel())
    d = float((a['det'].double() - b['det'].double()).norm() / b['det'].double().norm())
    flips = float(((a['det'] > 0) != (b['det'] > 0)).float().mean())
    print('%s: gradient arena rel %.3e, sign differences %.4f, losses %s' % (what, d, flips, {k: '%.6g' % v for k, v in a['losses'].items()}))
    assert d < grad_bar, (what, d)
    return d


def test_full_size_step_b32_four_cameras(monkeypatch):
    """BASELINE config 2 at full size: HM36_Multi_SurS1, 4 cameras, B = 32, one disc + gen step.  The camera-batched step
    (ONE grouped pass of 3 x 4 x 32 = 384 images) against the step that calls the networks once per camera: loss terms
    <= 1e-5 relative, depth-peak indices identical, gradient arena within 3e-2 in norm (measured 8e-3: the distance of two
    fp32 evaluations of this gradient), memory bound of the benchmark configuration."""
    a = _schedule_run('HM36_Multi_SurS1', 32, True, 'f16x3', monkeypatch)
    b = _schedule_run('HM36_Multi_SurS1', 32, False, 'f16x3', monkeypatch)
    # memory bound of the benchmark configuration: activations of the joint 12-group pass, both
    # weight-plane formats, gradient arenas and maxima - measured 67 GB batched (50 GB per camera); the part has 288
    assert a['peak_gb'] < 80 and b['peak_gb'] < 60, (a['peak_gb'], b['peak_gb'])
    _schedules_agree(a, b, 1e-5, 3e-2, 'S1 B=32 camera-batched vs per camera')


@pytest.mark.limit(300)
@pytest.mark.parametrize('name,batch', [('HM36_Multi_SurS2', 32), ('HM36_Multi_SynthS2', 64)])
def test_full_size_step_other_baseline_configs(name, batch, monkeypatch):
    """BASELINE configs 3 and 5 at the size BASELINE.json names (SurS2 finetune: symmetry + adversarial terms active,
    B = 32; SynthS2: B = 64 per GPU), one disc + gen step from a state with PLANTED depth peaks (the selections of
    modules/model.py:114,162 and keypoint_detector_integral_multi.py:24-34 are then not near-ties): camera-batched vs one
    call per camera, and the f16x3 default vs the exact-fp32 MFMA kernels: loss terms <= 1e-5 / 2e-5 relative, peak indices
    identical, gradient arenas within 3e-2 in norm."""
    a = _schedule_run(name, batch, True, 'f16x3', monkeypatch)
    b = _schedule_run(name, batch, False, 'f16x3', monkeypatch)
    c = _schedule_run(name, batch, True, 'f32', monkeypatch)
    assert any(v != 0.0 for k, v in a['losses'].items() if k in ('symmetry', 'smpl_gen'))          # the S2 terms are live
    _schedules_agree(a, b, 1e-5, 3e-2, '%s camera-batched vs per camera' % name)
    _schedules_agree(a, c, 2e-5, 3e-2, '%s f16x3 vs exact fp32' % name)
