"""Tiny detector forward + backward on the GPU vs the oracle, then one full optimisation step."""
import torch


def run():
    import inputs as gi
    from modules.keypoint_detector_integral_multi import KPDetector3DMulti
    from oracle import step as ostep
    from . import engine
    from .synthetic import model_config, synthetic_batch
    ora = gi.seeded_fill_(ostep.Regressor('resnet_multi', 18, 64, 3, 15), seed=61).train()
    with torch.no_grad():
        ora.net.head.features[9].bias.copy_(torch.from_numpy(gi.planted_depth_bias(18, 64, seed=62)))
    hip = KPDetector3DMulti('resnet_multi', 18, 64, 3, 15)
    hip.load_state_dict(ora.state_dict())
    hip.cuda().train()
    x = torch.from_numpy(gi.synthetic_batch(1, [0], seed=5)['cam_0_img'])
    x = torch.cat([x, x.flip(-1)])
    ko, _ = ora(x)
    kg, _ = hip(x.cuda())
    kg.sum().backward()
    torch.cuda.synchronize()
    err = (kg.detach().cpu() - ko.detach()).abs().max().item()
    assert err < 1e-4, err
    print('[smoke] detector ok: max|dkps| vs oracle = %.2e' % err)
    cfg = model_config('HM36_Multi_SurS1')
    cfg['model_params']['cam_id_list'] = [0]
    torch.manual_seed(0)                                   # reproducible weights: the printed losses are comparable between runs
    model, disc, od, odisc = engine.prepare_model(cfg)
    model.cuda().train(), disc.cuda().train()
    step = engine.TrainStep(cfg, model, disc, od, odisc)
    ld, lk, tot, _ = step(synthetic_batch(2, [0], torch.device('cuda'), seed=3))
    torch.cuda.synchronize()
    assert torch.isfinite(tot) and torch.isfinite(ld)
    print('[smoke] full step ok: loss_disc=%.4f total=%.4f' % (float(ld.detach()), float(tot.detach())))
