"""Two ranks on ONE GPU (gloo backend with device tensors) run the full data-parallel optimisation step:
side-stream bucketed gradient averaging, SyncBatchNorm statistic exchange, parameter broadcast.  After the
step the replicas must hold identical parameters and identical synchronised running statistics."""
import os

import pytest
import torch
import torch.distributed as dist

from _ranks import init_group, run_ranks

pytestmark = [pytest.mark.gpu, pytest.mark.multiproc]


def _worker(rank, world, opts=None):
    opts = opts or {}
    os.environ.update(XAS_DP_NOTIFY=str(opts.get('notify', 1)), XAS_CAM_BATCH=str(opts.get('cam_batch', 1)),
                      XAS_DP_OVERLAP=str(opts.get('overlap', 0)))
    os.environ.pop('XAS_DISC_BESIDE_GEN', None)               # unset = the data-parallel default (main stream, engine.py)
    if opts.get('beside') is not None:
        os.environ['XAS_DISC_BESIDE_GEN'] = str(opts['beside'])
    torch.cuda.set_device(0)
    init_group('gloo', rank, world)
    from xas_amd import engine
    from xas_amd.synthetic import model_config, synthetic_batch
    cfg = model_config('HM36_Multi_SurS2')
    cfg['model_params']['cam_id_list'] = opts.get('cams', [0])
    torch.manual_seed(100 + rank)                         # different init per rank: the broadcast must fix it
    model, disc, od, odisc = engine.prepare_model(cfg)
    model.cuda().train(), disc.cuda().train()
    disc.smpl_discriminator.header.p = 0.0
    step = engine.TrainStep(cfg, model, disc, od, odisc, num_buckets=3, dedupe=bool(opts.get('dedupe', False)))
    assert step.red_det is not None and len(step.red_det.buckets) >= 2
    early = []
    orig_launch = step.red_det._launch
    step.red_det._launch = lambda b: (early.append(step.red_det._armed), orig_launch(b))[1]
    x = synthetic_batch(2, opts.get('cams', [0]), torch.device('cuda'), seed=10 + rank)   # different data per rank
    ld, lk, tot, _ = step(x)
    torch.cuda.synchronize()
    p = od.param_arena
    sd = model.state_dict()
    res = (float(p.double().sum()), float(p.double().abs().sum()), float(odisc.param_arena.double().sum()),
                 float(sd['regressor.net.backbone.bn1.running_mean'].double().sum()),
                 float(sd['regressor.net.backbone.layer1.0.bn1.running_mean'].double().sum()),
                 bool(torch.isfinite(tot)), int(sum(early)))
    dist.destroy_process_group()          # (on an exception the harness exits the rank without waiting for its peers)
    return res


def _run2(opts=None):
    ret = run_ranks(_worker, 2, (opts,))
    return ret[0], ret[1]


@pytest.mark.parametrize('beside', [None, pytest.param(1, marks=pytest.mark.multistream), 0])
def test_two_rank_step_keeps_replicas_identical(beside):
    """beside = XAS_DISC_BESIDE_GEN: the discriminator update on the second stream next to the generator's detector passes
    (1, the single-GPU default) or on the main stream in program order (0; also what an unset variable means under data
    parallelism: None)."""
    a, b = _run2(dict(beside=beside))
    assert a[5] and b[5]
    assert a[0] == b[0] and a[1] == b[1]          # generator parameters bit-identical after the averaged step
    assert a[2] == b[2]                           # discriminator parameters too
    assert a[3] == b[3]                           # SyncBatchNorm (stem) running mean is global
    assert a[4] != b[4]                           # in-block BatchNorm2d stays rank-local (different data)


@pytest.mark.multistream
@pytest.mark.parametrize('dedupe,cam_batch', [(False, 1), (True, 1), (True, 0)])
def test_early_bucket_launch_equals_launch_at_finish(dedupe, cam_batch):
    """(XAS_DP_OVERLAP=1, the r02-r04 schedule - off by default since r05.)
    Buckets launched from the readiness reports during backward (XAS_DP_NOTIFY=1) must carry COMPLETE gradients: the
    step must be bit-identical to the one whose buckets are all launched by finish() (XAS_DP_NOTIFY=0), with two ranks
    (the all-reduce is not the identity), with and without TrainStep(dedupe=True) - whose real-image detector forward is
    counted before the discriminator step (r02 ADVICE: those counts were wiped and buckets left early, incomplete) - and
    with one detector call per camera (XAS_CAM_BATCH=0: several uses per parameter and pass)."""
    opts = dict(dedupe=dedupe, cam_batch=cam_batch, cams=[0, 1], overlap=1)
    a1, b1 = _run2(dict(opts, notify=1))
    a0, b0 = _run2(dict(opts, notify=0))
    assert a1[5] and a0[5]
    assert a1[:3] == b1[:3] and a0[:3] == b0[:3]               # replicas agree within each mode
    assert a1[:3] == a0[:3], (a1, a0)                          # and early launches change nothing


@pytest.mark.limit(240)
def test_bench_two_rank_rehearsal():
    """`python bench.py --gpus 2 --backend gloo` on ONE card: the launcher path the driver uses for N > 1 (a parent that never
    touches the GPU starts the ranks as a child torchrun job), the barrier / MAX-over-ranks timing, the whole-job line of rank 0:
    n_gpus == ranks == 2, value = both ranks' images, and the replicas' parameter checksums identical after the averaged steps."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('XAS_DISC_BESIDE_GEN', None)
    from _ranks import release_gpu_memory
    release_gpu_memory()                     # the child needs the card's memory, not this process's cache
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--batch', '4', '--steps', '2',
                        '--warmup', '1', '--f32-steps', '0', '--no-cpu-baseline', '--ref-n1', '100.0'], capture_output=True, text=True,
                       timeout=200, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['ranks'] == 2 and line['config']['parallelism'] == 'dp2'
    assert line['scaling'] == 'weak' and line['steps'] == 2 and line['warmup'] == 1
    assert abs(line['value'] - 2 * 4 * 8 * 2 / (line['ms_per_step'] * 2e-3)) < 1e-6 * line['value']      # whole-job images / time
    assert line['config']['replicas_identical'] is True, line['config']['param_checksum_per_rank']
    # the self-explaining multi-GPU line: what the step sends and what it waits for
    c = line['comm']
    for k in ('gradient_buckets', 'gradient_MB_per_step', 'gradient_wait_compute_stream_ms', 'gradient_wait_host_ms',
              'syncbn_exchanges_per_step', 'syncbn_KB_per_exchange', 'syncbn_host_ms', 'per_rank_over_n1', 'step_ms'):
        assert k in c, k
    names = [b['reducer'] for b in c['gradient_buckets']]
    assert names.count('detector') == 4 and names.count('discriminator') == 1, names          # 4 + 1 buckets (train.py:87-88: two DDP wrappers)
    assert 130 < c['gradient_MB_per_step'] < 160                                             # 138.8 MB + 10.8 MB of gradients
    assert c['syncbn_exchanges_per_step'] > 0 and c['syncbn_KB_per_exchange'] > 0
    assert abs(c['per_rank_over_n1'] - line['config']['samples_per_s_per_rank'] / 100.0) < 1e-9
