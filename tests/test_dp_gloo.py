"""Data-parallel path on CPU with the gloo backend, world_size 2: the bucketed gradient reducer averages
exactly like a single process on the concatenated batch, SyncBatchNorm statistic exchange reproduces the
global-batch statistics, buffers are broadcast from rank 0.  (No GPU, no HIP kernels: the reducer and the
statistic exchange are backend-agnostic torch.distributed code.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist

from _ranks import init_group, run_ranks

pytestmark = pytest.mark.multiproc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, fn):
    torch.set_num_threads(2)
    init_group('gloo', rank, world)
    res = fn(rank, world)
    dist.destroy_process_group()
    return res


def _run(fn, world=2):
    return run_ranks(_worker, world, (fn,), limit=100)


def _reducer_case(rank, world):
    from xas_amd.dp import GradReducer
    torch.manual_seed(0)                                   # identical parameters on every rank
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                              torch.nn.Linear(16, 4))
    params = list(net.parameters())
    offs, n = [], 0
    for p in params:
        offs.append(n)
        n += (p.numel() + 3) // 4 * 4
    arena = torch.zeros(n)
    for p, o in zip(params, offs):
        p.grad = arena[o:o + p.numel()].view(p.shape)      # gradients accumulate into the flat arena
    red = GradReducer(arena, params, offs, num_buckets=3, use_side_stream=False)
    assert red.enabled and len(red.buckets) >= 2
    g = torch.Generator().manual_seed(123)
    x_all = torch.randn(world * 6, 8, generator=g)
    x = x_all[rank * 6:(rank + 1) * 6]
    out = []
    for it in range(2):                                    # two steps: hooks re-arm correctly
        arena.zero_()
        red.arm()
        net(x).pow(2).mean().backward()
        red.finish()
        out.append(arena.clone())
    # reference: single process, mean over ranks of the per-rank mean losses
    ref = torch.zeros(n)
    for r in range(world):
        net2 = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                                   torch.nn.Linear(16, 4))
        net2.load_state_dict(net.state_dict())
        net2(x_all[r * 6:(r + 1) * 6]).pow(2).mean().backward()
        for p, o in zip(net2.parameters(), offs):
            ref[o:o + p.numel()] += p.grad.reshape(-1) / world
    return [float((o - ref).abs().max()) for o in out]


def _reducer_case_overlap(rank, world):
    import os
    os.environ['XAS_DP_OVERLAP'] = '1'              # buckets leave from the hooks, during backward (the r02-r04 schedule)
    return _reducer_case(rank, world)


@pytest.mark.parametrize('case', [_reducer_case, _reducer_case_overlap], ids=['at_finish', 'overlap'])
def test_bucketed_reducer_world2(case):
    res = _run(case)
    for r in (0, 1):
        assert max(res[r]) < 1e-6


def _syncbn_stats_case(rank, world):
    from xas_amd.ops_nn import _sync_stats
    g = torch.Generator().manual_seed(7)
    x_all = torch.randn(world * 10, 5, generator=g) * 3 + 1
    x = x_all[rank * 10:(rank + 1) * 10]
    mean, var = x.mean(0), x.var(0, unbiased=False)
    gm, gv = _sync_stats(mean, var, x.shape[0], dist.group.WORLD)
    return float((gm - x_all.mean(0)).abs().max()), float((gv - x_all.var(0, unbiased=False)).abs().max())


def test_syncbn_statistic_exchange_world2():
    res = _run(_syncbn_stats_case)
    for r in (0, 1):
        assert res[r][0] < 1e-6 and res[r][1] < 1e-5


def _buffers_case(rank, world):
    from xas_amd.dp import sync_buffers
    bn = torch.nn.BatchNorm1d(6)
    with torch.no_grad():
        bn.running_mean.fill_(float(rank + 1))
        bn.running_var.fill_(float(10 * (rank + 1)))
    sync_buffers(bn)
    return float(bn.running_mean[0]), float(bn.running_var[0]), int(bn.num_batches_tracked)


def test_buffer_broadcast_world2():
    res = _run(_buffers_case)
    assert res[0][:2] == (1.0, 10.0) and res[1][:2] == (1.0, 10.0)


def test_reducer_disabled_without_process_group():
    sys.path.insert(0, os.path.join(ROOT, 'x-as-supervision_amd'))
    from xas_amd.dp import GradReducer
    p = torch.nn.Parameter(torch.zeros(4))
    red = GradReducer(torch.zeros(4), [p], [0])
    assert not red.enabled
    red.arm()
    red.finish()


def test_use_counts_are_per_reducer():
    """r02 ADVICE (medium): under TrainStep(dedupe=True) the detector's real-image forward is counted BEFORE the
    discriminator step; the discriminator reducer's finish() must not wipe those counts, and a parameter without a
    counted use must never be reported early (its bucket is left to finish())."""
    sys.path.insert(0, os.path.join(ROOT, 'x-as-supervision_amd'))
    from xas_amd import ops_nn
    det = [torch.nn.Parameter(torch.zeros(4)) for _ in range(2)]
    disc = [torch.nn.Parameter(torch.zeros(4))]
    fired = []
    ops_nn.track_grad_uses(True)
    try:
        ops_nn._uses['hook'] = lambda p: fired.append(p.data_ptr())
        for p in det:                       # real-image detector pass, counted first (dedupe)
            ops_nn.note_use(p)
        ops_nn.note_use(disc[0])            # discriminator step
        ops_nn.grad_ready(disc[0])
        assert fired == [disc[0].data_ptr()]
        ops_nn.forget_uses([disc[0].data_ptr()])          # what red_disc.finish() does now
        assert all(p.data_ptr() in ops_nn._uses['pending'] for p in det)
        for p in det:                       # pseudo-image pass: second use of every detector parameter
            ops_nn.note_use(p)
        fired.clear()
        ops_nn.grad_ready(det[0])           # first of two contributions: not complete
        assert fired == []
        ops_nn.grad_ready(det[0])           # second: complete
        assert fired == [det[0].data_ptr()]
        ops_nn.grad_ready(det[0])           # a contribution nobody counted: never reported
        assert fired == [det[0].data_ptr()]
        stranger = torch.nn.Parameter(torch.zeros(4))
        ops_nn.grad_ready(stranger)
        assert fired == [det[0].data_ptr()]
    finally:
        ops_nn._uses['hook'] = None
        ops_nn.track_grad_uses(False)
