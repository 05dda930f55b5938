"""The RCCL (`nccl` backend) code path of the data-parallel step on ONE GPU: a world-size-1 nccl process group with
XAS_FORCE_DP=1 drives every collective call of the path through RCCL - parameter broadcast, packed SyncBatchNorm
all-gather + merge kernel, backward all-reduce, bucketed gradient all-reduce (own communicator; by default launched by
finish() after backward, with XAS_DP_OVERLAP=1 from the readiness hooks during backward on the communication stream).  With
one rank the exchange must be the identity: the step must equal the step without a process group."""
import os

import pytest
import torch

from _ranks import init_group, run_ranks

pytestmark = [pytest.mark.gpu, pytest.mark.multiproc]


def _worker(rank, world, mode):
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if mode.startswith('nccl'):
        os.environ['XAS_FORCE_DP'] = '1'
        os.environ['XAS_DP_OVERLAP'] = '1' if mode == 'nccl_overlap' else '0'
        mode = 'nccl'
        init_group('nccl', 0, 1, device_id=torch.device('cuda', 0))
    from xas_amd import engine
    from xas_amd.synthetic import model_config, synthetic_batch
    cfg = model_config('HM36_Multi_SurS2')
    cfg['model_params']['cam_id_list'] = [0, 1]
    torch.manual_seed(7)
    model, disc, od, odisc = engine.prepare_model(cfg)
    model.cuda().train(), disc.cuda().train()
    disc.smpl_discriminator.header.p = 0.0
    step = engine.TrainStep(cfg, model, disc, od, odisc, num_buckets=3)
    if mode == 'nccl':
        assert step.red_det is not None and step.red_det.enabled and len(step.red_det.buckets) >= 2
        assert dist.get_backend(step.red_det.group) == 'nccl'
        launched = []
        orig = step.red_det._launch
        step.red_det._launch = lambda b: (launched.append(step.red_det._armed), orig(b))[1]
    x = synthetic_batch(2, [0, 1], torch.device('cuda'), seed=21)
    losses = []
    for _ in range(2):
        ld, lk, tot, _o = step(x)
        losses.append((float(ld.detach()), float(tot.detach())))
    torch.cuda.synchronize()
    sd = model.state_dict()
    res = (losses, od.param_arena.double().sum().item(), od.param_arena.double().abs().sum().item(),
                 odisc.param_arena.double().sum().item(), sd['regressor.net.backbone.bn1.running_mean'].double().sum().item(),
                 sd['regressor.net.backbone.bn1.running_var'].double().sum().item(),
                 (sum(launched), len(launched)) if mode == 'nccl' else None)
    if mode == 'nccl':
        dist.destroy_process_group()
    return res


@pytest.mark.parametrize('mode', ['nccl', pytest.param('nccl_overlap', marks=pytest.mark.multistream)])
def test_nccl_world1_step_equals_plain_step(mode):
    a = run_ranks(_worker, 1, ('plain',))[0]
    b = run_ranks(_worker, 1, (mode,))[0]
    early, total = b[6]
    assert total == 2 * 3                                    # 3 buckets per generator step
    if mode == 'nccl_overlap':
        assert early >= 1                                    # some launched from the hooks, during backward
    else:
        assert early == 0                                    # default: nothing travels beside backward
    for (la, ta), (lb, tb) in zip(a[0], b[0]):
        assert abs(la - lb) <= 1e-6 * max(1.0, abs(la)) and abs(ta - tb) <= 1e-5 * max(1.0, abs(ta))
    # identity exchange: parameters and synchronised running statistics agree with the plain step (the merge kernel
    # re-derives mean / var from the gathered message: same values up to one rounding)
    assert abs(a[1] - b[1]) <= 1e-6 * abs(a[2]) and abs(a[3] - b[3]) <= 1e-9 + 1e-6 * abs(a[3])
    assert abs(a[4] - b[4]) <= 1e-6 * max(1.0, abs(a[4])) and abs(a[5] - b[5]) <= 1e-6 * max(1.0, abs(a[5]))
