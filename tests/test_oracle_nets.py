"""Oracle detector / model wiring vs goldens from the imported reference.  CPU only."""
import numpy as np
import torch
import torch.nn as nn

import inputs as gi
from conftest import golden
from oracle import step as ostep

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, atol, rtol=0.0):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), atol=atol, rtol=rtol)


def make_regressor(multi=True):
    if multi:
        reg = ostep.Regressor('resnet_multi', 18, 64, 3, 15)
    else:
        reg = ostep.Regressor('resnet', 18, 64)
    gi.seeded_fill_(reg, seed=61)
    with torch.no_grad():
        reg.net.head.features[9].bias.copy_(T(gi.planted_depth_bias(18, 64, seed=62)))
    return reg


def test_detector_end_to_end():
    g = golden('detector')
    reg = make_regressor()
    sd = reg.state_dict()
    assert list(sd.keys()) == g['keys'].tolist()               # checkpoint key names are API
    assert [str(list(v.shape)) for v in sd.values()] == g['shapes'].tolist()
    assert sum(p.numel() for p in reg.parameters()) == 34291392
    reg.train()
    x = T(gi.synthetic_batch(2, [0], seed=63)['cam_0_img'])
    heat = reg.net(x)
    close(heat[:, ::37, ::4, ::4], g['heat_sub'], 2e-4, 1e-4)
    reg2 = make_regressor()
    reg2.train()
    kps, dmap = reg2(x)
    close(kps, g['kps'], 1e-4)                                  # north-star bar, normalised space
    close(dmap, g['depth_prob_map'], 1e-5)
    (kps * T(g['grad_out'])).sum().backward()
    p = dict(reg2.named_parameters())
    rel = lambda a, b: float((a - T(b)).norm() / (T(b).norm() + 1e-30))
    assert rel(p['net.backbone.conv1.weight'].grad, g['g_conv1']) < 2e-3
    assert rel(p['net.backbone.layer1.0.conv2.weight'].grad[:8], g['g_l1c2']) < 2e-3
    assert rel(p['net.head.features.9.bias'].grad, g['g_fin_b']) < 1e-3
    assert abs(p['net.backbone.layer4.2.conv3.weight'].grad.norm().item() / g['g_l4c3_norm'] - 1) < 1e-3
    close(reg2.state_dict()['net.backbone.bn1.running_mean'], g['rm_bn1'], 1e-6)
    close(ostep.Regressor.forward(make_regressor(False).train(), x)[0], golden('detector_single')['kps'], 1e-4)


class LinearDisc(nn.Module):
    name = 'LinearStandIn'

    def __init__(self):
        super().__init__()
        self.fc = nn.Linear(54, 1)

    def forward(self, kp):
        return self.fc(kp.reshape(kp.shape[0], -1))


def _run_model(stage, gname=None, cfg=None, cams=(0, 1), seed=83, rng_seed=None):
    from oracle.nets import PhysiqueNet
    g = golden(gname or 'model_HM36_Multi_Sur' + stage)
    cfg = cfg or gi.model_params(stage, cam_ids=(0, 1))
    reg = make_regressor().train()
    phys = gi.seeded_fill_(PhysiqueNet([32, 64, 128]), seed=81).train()
    disc = gi.seeded_fill_(LinearDisc(), seed=82)
    x = {k: T(v) for k, v in gi.synthetic_batch(2, list(cams), seed=seed).items()}
    if rng_seed is not None:
        torch.manual_seed(rng_seed)
    ld = ostep.discriminator_loss(cfg, reg, disc, x)
    close(ld, g['loss_disc'], 1e-5, 1e-4)
    ld.mean().backward()
    close(disc.fc.weight.grad, g['grad_disc_w'], 1e-5, 1e-3)
    disc.zero_grad()
    losses, aux = ostep.generator_losses(cfg, reg, phys, disc, x, return_aux=True)
    for k, v in losses.items():
        assert list(v.shape) == g['shape_' + k].tolist(), k
        close(v.mean(), g['loss_' + k], 1e-5, 2e-4)
    tot = sum(v.mean() for v in losses.values())
    close(tot, g['total'], 1e-5, 2e-4)
    tot.backward()
    close(aux['world']['cam_%d' % cams[0]][:, 0], g['pose_3d_cam_0'], 0.05, 1e-5)
    close(aux['recon']['cam_%d' % cams[-1]][:, :, ::4, ::4], g['mask_line_sub'], 1e-4)
    p = dict(reg.named_parameters())
    rel = lambda a, b: float((a - T(b)).norm() / (T(b).norm() + 1e-30))
    assert rel(p['net.backbone.conv1.weight'].grad, g['g_conv1']) < 5e-3
    assert rel(p['net.head.features.9.bias'].grad, g['g_fin_b']) < 2e-3
    assert rel(phys.decoder[4].weight.grad, g['g_phys_dec4_w']) < 2e-3 or float(T(g['g_phys_dec4_w']).norm()) == 0
    if float(T(g['g_disc_after_gen']).norm()) > 0:
        assert rel(disc.fc.weight.grad, g['g_disc_after_gen']) < 2e-3


def test_model_wiring_s1():
    _run_model('S1')


def test_model_wiring_s2():
    _run_model('S2')


def _yaml_params(name, cams):
    from xas_amd.synthetic import model_config           # pure Python, equal to the YAML (tests/test_configs.py)
    mp = model_config(name)['model_params']
    mp['cam_id_list'] = list(cams)
    return mp


def test_model_wiring_s1_weighted_mask_losses():
    """S1 with non-zero mask-loss weights and use_dis_map: True (the shipped 0.0 hides those terms)."""
    mp = _yaml_params('HM36_Multi_SurS1', (0, 1))
    mp['loss_config']['recons_loss']['weight'] = 0.02
    mp['loss_config']['physique_recons_loss']['weight'] = 0.02
    _run_model(None, 'model_HM36_Multi_SurS1_wmask', mp)


def test_model_wiring_synth_s2():
    _run_model(None, 'model_HM36_Multi_SynthS2', _yaml_params('HM36_Multi_SynthS2', (0, 1)), seed=85)


def test_model_wiring_mpi_five_cameras():
    cams = (0, 2, 4, 7, 8)
    _run_model(None, 'model_MPI_Multi_SurS1', _yaml_params('MPI_Multi_SurS1', cams), cams, seed=84)


def test_model_wiring_use_aug():
    """use_aug: random z-rotations (util.py:389-407) with the CPU generator seeded as in the golden run; the rotated
    generator branch is not detached (model.py:136), so the detector receives an adversarial gradient here."""
    mp = _yaml_params('HM36_Multi_SurS2', (0, 1))
    mp['smpl_disc_params']['use_aug'] = True
    _run_model(None, 'model_HM36_Multi_SurS2_aug', mp, seed=86, rng_seed=1234)


def test_model_wiring_s1_four_cameras():
    """HM36_Multi_SurS1 with the YAML's four cameras and the mask losses switched on (golden: model4)."""
    cams = (0, 1, 2, 3)
    mp = _yaml_params('HM36_Multi_SurS1', cams)
    mp['loss_config']['recons_loss']['weight'] = 0.02
    mp['loss_config']['physique_recons_loss']['weight'] = 0.02
    _run_model(None, 'model_HM36_Multi_SurS1_4cam', mp, cams, seed=87)


def test_detector_all_parameter_gradients():
    """Every one of the detector's 170 parameter gradients (norm + strided sample) against the imported reference, on the
    smooth single-hypothesis head without the planted depth bias (golden: detector_allgrads)."""
    g = golden('detector_allgrads')
    reg = gi.seeded_fill_(ostep.Regressor('resnet', 18, 64), seed=61).train()
    x = T(gi.synthetic_batch(2, [0], seed=63)['cam_0_img'])
    kps, _ = reg(x)
    close(kps, g['kps'], 1e-5)
    gw = T(np.random.Generator(np.random.PCG64(64)).standard_normal(kps.shape).astype(np.float32))
    (kps * gw).sum().backward()
    from conftest import check_all_grads
    check_all_grads([(n, p.grad) for n, p in reg.named_parameters()], g, 2e-3, 2.0, 'oracle')
