"""Two ranks on ONE GPU (gloo backend with device tensors) run the full data-parallel optimisation step:
side-stream bucketed gradient averaging, SyncBatchNorm statistic exchange, parameter broadcast.  After the
step the replicas must hold identical parameters and identical synchronised running statistics."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0')
    for p in (ROOT, os.path.join(ROOT, 'x-as-supervision_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from xas_amd import engine
        from xas_amd.synthetic import model_config, synthetic_batch
        cfg = model_config('HM36_Multi_SurS2')
        cfg['model_params']['cam_id_list'] = [0]
        torch.manual_seed(100 + rank)                         # different init per rank: the broadcast must fix it
        model, disc, od, odisc = engine.prepare_model(cfg)
        model.cuda().train(), disc.cuda().train()
        disc.smpl_discriminator.header.p = 0.0
        step = engine.TrainStep(cfg, model, disc, od, odisc, num_buckets=3)
        assert step.red_det is not None and len(step.red_det.buckets) >= 2
        x = synthetic_batch(2, [0], torch.device('cuda'), seed=10 + rank)   # different data per rank
        ld, lk, tot, _ = step(x)
        torch.cuda.synchronize()
        p = od.param_arena
        sd = model.state_dict()
        ret[rank] = (float(p.double().sum()), float(p.double().abs().sum()), float(odisc.param_arena.double().sum()),
                     float(sd['regressor.net.backbone.bn1.running_mean'].double().sum()),
                     float(sd['regressor.net.backbone.layer1.0.bn1.running_mean'].double().sum()),
                     bool(torch.isfinite(tot)))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_keeps_replicas_identical():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    a, b = ret[0], ret[1]
    assert a[5] and b[5]
    assert a[0] == b[0] and a[1] == b[1]          # generator parameters bit-identical after the averaged step
    assert a[2] == b[2]                           # discriminator parameters too
    assert a[3] == b[3]                           # SyncBatchNorm (stem) running mean is global
    assert a[4] != b[4]                           # in-block BatchNorm2d stays rank-local (different data)
