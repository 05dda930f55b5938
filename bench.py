#!/usr/bin/env python3
"""Benchmark of the MI355X-native X-as-Supervision training step (BASELINE.json north_star).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = the reference's hot-loop body (train.py:160-190): discriminator forward/backward/Adam, then
generator forward/backward/Adam, on one synthetic batch of the workload HM36_Multi_SurS1 (4 cameras,
256x256, multi-hypothesis detector), B = 32 samples per GPU, fp32, inputs resident in HBM.  One sample
= 4 real + 4 pseudo images = 8 input images; the step runs 12 detector forwards and 8 detector backwards
per sample.  Rank 0 prints ONE JSON line (metric = images/s over the whole job).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'x-as-supervision_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this driver: RCCL needs it (multi-process runs)
import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, 'Peak FP32 (matrix)': v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 16 * PEAK_FP32_MFMA_TFLOPS      # dense bf16 MFMA = 16 x the fp32 MFMA rate (same table): 2516.8
# fp32-equivalent peak of a bf16x6 launch: six bf16 MFMAs per fp32 product (VERDICT r02 ruling): 419.5 TFLOP/s
# f16x3: three fp16 MFMAs (same rate as bf16) per fp32 product: 2516.8 / 3 = 838.9 TFLOP/s fp32-equivalent
PEAK_BY_CLASS = {':f32': PEAK_FP32_MFMA_TFLOPS, ':bf16': PEAK_BF16_MFMA_TFLOPS, ':bf16x6': PEAK_BF16_MFMA_TFLOPS / 6.0,
                 ':f16x3': PEAK_BF16_MFMA_TFLOPS / 3.0}
IMAGES_PER_SAMPLE = {'HM36': 8, 'MPI': 10}


def host_threads():
    """Cores this process may use: the GPU box gives one GPU's share (16), not the host's core count."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(workload, threads, B=2, warm=3, timed=5):
    """Oracle (CPU restatement of the reference step, stock PyTorch fp32) on a bounded sample: full disc+gen steps on the host
    cores.  BASELINE.md section 3 protocol: B = 2 (config 1's batch), 3 warm-up + 5 timed steps, median samples/s (reported as
    images/s like the headline); bench.py also times B = 8 (1 + 3) as `cpu_baseline_b8`, the r01-r04 figure."""
    from oracle import step as ostep
    from oracle.nets import GCNDecouple, PhysiqueNet
    from xas_amd.synthetic import model_config, synthetic_batch
    torch.set_num_threads(threads)
    cfg = model_config(workload)['model_params']
    torch.manual_seed(0)
    reg = ostep.Regressor(**cfg['detector_params']).train()
    phys = PhysiqueNet(cfg['physique_mask_generator_params']['layers']).train()
    disc = GCNDecouple(cfg['smpl_disc_params'])
    from oracle.geometry import skeleton_links
    disc.parent_ids, disc.child_ids = skeleton_links(cfg['parent_ids'], cfg['line_select_ids'], False, False)
    o_det = torch.optim.Adam(list(reg.parameters()) + list(phys.parameters()), lr=2e-4, betas=(0.5, 0.999))
    o_disc = torch.optim.Adam(disc.parameters(), lr=2e-4, betas=(0.5, 0.999))
    x = synthetic_batch(B, cfg['cam_id_list'], torch.device('cpu'), seed=1)
    times = []
    for i in range(warm + timed):
        t0 = time.perf_counter()
        ostep.train_step(cfg, reg, phys, disc, o_det, o_disc, x)
        times.append(time.perf_counter() - t0)
    ts = sorted(times[warm:])
    dt = ts[len(ts) // 2]
    per_sample = IMAGES_PER_SAMPLE['MPI' if workload.startswith('MPI') else 'HM36']
    return {'value': B * per_sample / dt, 'unit': 'images/s', 'samples_per_s': B / dt, 'cores': threads, 'cpu': cpu_model_name(),
            'kind': 'port', 'protocol': '%d warm-up + %d timed steps, median (BASELINE.md section 3)' % (warm, timed),
            'sample': 'full disc+gen steps incl. Adam, %s, B=%d, fp32, oracle on torch CPU (warm-up %s s, timed %s s)'
                      % (workload, B, ' '.join('%.1f' % t for t in times[:warm]), ' '.join('%.1f' % t for t in times[warm:]))}


def train_step_variant_check(step, x, model, disc, opt_det, opt_disc, xl, precision, base_tune=0):
    """One full TRAIN-MODE step at the benchmark size in `precision` and one on the exact-fp32 MFMA kernels, both from the SAME
    state (parameters, Adam moments, running statistics snapshot / restored): relative difference of every loss term and of
    the two gradient arenas as the optimizers would consume them (condition c)."""
    import torch
    f_det, f_disc = opt_det._flat, (opt_disc._flat if opt_disc is not None else None)
    bufs = [b for m in (model, disc) for b in m.buffers()]

    def snapshot():
        # (the gradient arenas too: the generator's backward leaves gradients in the DISCRIMINATOR's arena that the next
        # discriminator update consumes - train.py:160-190 never zeroes them in between, and neither does the mirror)
        return ([t.clone() for f in (f_det, f_disc) if f is not None for t in (f['p'], f['m'], f['v'], f['g'])],
                [b.clone() for b in bufs], opt_det._steps, opt_disc._steps if opt_disc is not None else 0, step.cur_step)

    def restore(sn):
        ts, bs, s_det, s_disc, cur = sn
        it = iter(ts)
        with torch.no_grad():
            for f in (f_det, f_disc):
                if f is not None:
                    for k in ('p', 'm', 'v', 'g'):
                        f[k].copy_(next(it))
            for b, v in zip(bufs, bs):
                b.copy_(v)
        opt_det._steps = s_det
        opt_det._epoch[0] += 1                       # packed weight copies are stale
        if opt_disc is not None:
            opt_disc._steps = s_disc
            opt_disc._epoch[0] += 1
        step.cur_step = cur

    def run(prec, tune=None):
        tune = base_tune if tune is None else tune
        xl.query('xas_set_precision', prec)
        xl.query('xas_set_tuning', tune)
        torch.manual_seed(4242)                      # the discriminator's dropout mask: the same in both runs
        grads = {}
        step.grad_probe = lambda which, arena: grads.__setitem__(which, arena.clone())
        ld, lk, tot, _ = step(x)
        step.grad_probe = None
        torch.cuda.synchronize()
        xl.query('xas_set_tuning', base_tune)
        losses = {'disc': float(ld.detach()) if ld is not None else 0.0, 'total': float(tot.detach())}
        losses.update({k: float(v.detach().mean()) for k, v in lk.items()})
        return losses, grads

    sn = snapshot()
    la, ga = run(xl.PREC_NAMES[precision])
    restore(sn)
    lb, gb = run(xl.PREC_F32)
    restore(sn)
    ref = None
    if precision != 'bf16x6':                        # the range-free six-product mode against the same exact-fp32 run: the yardstick
        lc, gc = run(xl.PREC_BF16X6)
        restore(sn)
        ref = {'max_loss_rel_diff': max([abs(lc[k] - lb[k]) / max(abs(lb[k]), 1e-30) for k in lb if lb[k] != 0.0 or lc[k] != 0.0] or [0.0]),
               'grad_arena_rel_diff': {k: float((gc[k] - gb[k]).double().norm() / gb[k].double().norm().clamp_min(1e-300)) for k in gb}}
    # ... and the exact-fp32 kernels against THEMSELVES with another summation order in the batch-norm column reductions
    # (tuning value 65536: 128 instead of 256 row slabs per reduction - statistics of 17 layers, the two backward sums of every
    # layer): how far two exact-fp32 evaluations of this step's gradient are apart
    ld_, gd_ = run(xl.PREC_F32, tune=65536)
    restore(sn)
    reorder = {'max_loss_rel_diff': max([abs(ld_[k] - lb[k]) / max(abs(lb[k]), 1e-30) for k in lb if lb[k] != 0.0 or ld_[k] != 0.0] or [0.0]),
               'grad_arena_rel_diff': {k: float((gd_[k] - gb[k]).double().norm() / gb[k].double().norm().clamp_min(1e-300)) for k in gb}}
    xl.query('xas_set_precision', xl.PREC_NAMES[precision])
    rel = {k: abs(la[k] - lb[k]) / max(abs(lb[k]), 1e-30) for k in lb if lb[k] != 0.0 or la[k] != 0.0}
    gr = {k: float((ga[k] - gb[k]).double().norm() / gb[k].double().norm().clamp_min(1e-300)) for k in gb}
    return {**({'bf16x6_vs_exact_fp32_same_state': ref} if ref else {}),
            'exact_fp32_other_summation_order_vs_exact_fp32_same_state': reorder,'what': 'ONE train-mode step (disc + gen, B as timed) in this precision mode vs the exact-fp32 MFMA kernels from the same '
                    'state: |loss_mode - loss_f32| / |loss_f32| per loss term, ||g_mode - g_f32|| / ||g_f32|| over each gradient arena '
                    'as handed to Adam',
            'loss_rel_diff': rel, 'max_loss_rel_diff': max(rel.values()) if rel else 0.0,
            'grad_arena_rel_diff': gr, 'grad_arena_norm_f32': {k: float(v.double().norm()) for k, v in gb.items()}}


def launch_ranks(n):
    """Run this script as n ranks of a single-node torchrun job (child process; no exec, no GPU use here)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=32, help='samples per GPU')
    ap.add_argument('--workload', default='HM36_Multi_SurS1')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo to rehearse ranks on one GPU)')
    ap.add_argument('--tune', type=int, default=0, help='kernel tuning experiment flags (xas_set_tuning)')
    ap.add_argument('--precision', default='f16x3', choices=['bf16x6', 'f16x3', 'f32', 'bf16'],
                    help='arithmetic of the MFMA convolutions (xas_hip.h XAS_PREC_*).  f16x3 (default, the headline; the '
                         'whole parity suite runs in this mode): every fp32 operand is split into two fp16 '
                         'pieces (22 bits, power-of-two scaled) and three partial products are accumulated in fp32 (peak 2516.8 / 3 = '
                         '838.9 TFLOP/s fp32-equivalent).  bf16x6: every pass with fp32 '
                         'operands split exactly into three bf16 pieces, six exact partial products accumulated in fp32 - '
                         'fp32-accurate, peak 2516.8 / 6 = 419.5 TFLOP/s fp32-'
                         'equivalent.  f32: exact fp32 MFMA (v_mfma_f32_32x32x2_f32, peak 157.3).  bf16: operands rounded '
                         'once - NOT fp32 accurate (misses the 1e-4 joint bar), a variant that is reported separately, never '
                         'the headline; peak 2516.8')
    ap.add_argument('--f32-steps', type=int, default=5,
                    help='extra UNTIMED-by-the-headline steps on the exact-fp32 MFMA kernels after the timed region, to print '
                         'that figure beside the bf16x6 headline (0 = skip)')
    ap.add_argument('--no-variant-check', action='store_true',
                    help='skip the untimed precision cross-checks after the timed region (profiling runs: keeps other modes out of the trace)')
    ap.add_argument('--cse-steps', type=int, default=5, help='extra UNTIMED-by-the-headline steps with TrainStep(dedupe=True) (0: skip)')
    ap.add_argument('--dedupe', action='store_true',
                    help='NOT the headline: share the real-image detector forward between the discriminator and the '
                         'generator update (engine.TrainStep(dedupe=True)); the JSON line is marked config.dedupe')
    ap.add_argument('--ref-n1', type=float, default=None,
                    help='samples/s of the SAME build at N = 1 (a previous run of this script): adds comm.per_rank_over_n1 to the line')
    ap.add_argument('--shape-report', default=None, help='write a per-conv-shape timing table to this file')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: this parent has not touched the GPU; it starts N fresh ranks (one per GPU, RCCL)
        # as a child torchrun job and exits with its code.  Rank 0 of the child prints the JSON line.
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d; launch with --nproc-per-node %d (or drop WORLD_SIZE and '
                         'let bench.py start the ranks itself)' % (args.gpus, world, args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    local = local % max(1, torch.cuda.device_count())     # rehearsal: several ranks on one card
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        import datetime
        limit = datetime.timedelta(minutes=10)       # a rank that never arrives fails the job instead of hanging it
        if args.backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend=args.backend, timeout=limit)

    from xas_amd import engine
    from xas_amd.prof import KernelTimer
    from xas_amd.synthetic import model_config, synthetic_batch
    from xas_amd import _lib as _xl
    _xl.query('xas_set_tuning', args.tune)
    _xl.query('xas_set_precision', _xl.PREC_NAMES[args.precision])
    cfg = model_config(args.workload)
    torch.manual_seed(1234)
    model, disc, opt_det, opt_disc = engine.prepare_model(cfg)
    model.to(dev).train()
    disc.to(dev).train()
    step = engine.TrainStep(cfg, model, disc, opt_det, opt_disc, dedupe=args.dedupe)
    cams = cfg['model_params']['cam_id_list']
    x = synthetic_batch(args.batch, cams, dev, seed=100 + rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print('[bench] ' + msg, file=sys.stderr, flush=True)

    log('model built, batch resident; warmup %d step(s)' % args.warmup)
    for i in range(args.warmup):
        tw = time.perf_counter()
        step(x)
        torch.cuda.synchronize()
        log('warmup step %d: %.1f ms' % (i, (time.perf_counter() - tw) * 1e3))
    sync()
    # HIP events bracket every conv-family launch of the LAST timed step (1 818 launches): bracketing all K steps cost
    # ~4 % of the headline (two event packets per launch on the queue), one step costs < 1 %.
    from xas_amd.prof import CONV_ENTRIES
    HEAD = ('xas_head_softargmax_fwd', 'xas_head_softargmax_from_partials', 'xas_head_softargmax_bwd', 'xas_head_softargmax_bwd_amax')
    from xas_amd.prof import BN_ENTRIES
    timer = KernelTimer(CONV_ENTRIES + HEAD + BN_ENTRIES)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1:
            with timer:
                step(x)
        else:
            step(x)
    sync()
    timed_steps = 1
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    log('timed %d steps: %.1f ms/step' % (args.steps, dt / args.steps * 1e3))
    log('memory: peak allocated %.1f GB, reserved %.1f GB' % (torch.cuda.max_memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))

    # N > 1: one extra, UNTIMED step with the data-parallel exchanges instrumented (xas_amd/dp.py comm_stats): what the step waits
    # for - the exposed part of the gradient buckets, the host-visible time of the SyncBatchNorm exchanges - and what it sends
    comm = None
    if world > 1:
        from xas_amd import dp as _dp
        _dp.comm_begin()
        tc0 = time.perf_counter()
        step(x)
        torch.cuda.synchronize()
        comm_ms = (time.perf_counter() - tc0) * 1e3
        st = _dp.comm_end()
        if rank == 0:
            nb = st['buckets']
            comm = {'what': 'ONE extra untimed step with the exchanges instrumented, rank 0',
                    'step_ms': comm_ms,
                    'gradient_buckets': [{'reducer': n_, 'MB': b_ / 1e6, 'launched_during_backward': e_} for n_, b_, e_ in nb],
                    'gradient_MB_per_step': sum(b_ for _, b_, _ in nb) / 1e6,
                    'gradient_wait_compute_stream_ms': st['grad_wait_stream_ms'],
                    'gradient_wait_host_ms': st['grad_wait_host_ms'],
                    'syncbn_exchanges_per_step': st['syncbn_calls'], 'syncbn_KB_per_exchange': (st['syncbn_bytes'] / max(1, st['syncbn_calls'])) / 1e3,
                    'syncbn_host_ms': st['syncbn_host_ms'],
                    'backend': args.backend, 'overlap_with_backward': _dp.overlap_enabled(),
                    'note': 'gradient_wait_* = the EXPOSED part of the bucket all-reduces: by default (XAS_DP_OVERLAP=0) the buckets '
                            'leave after backward with the compute stream waiting, so it is the whole exchange (DESIGN section 5: no '
                            'kernel runs beside this library\'s MFMA kernels); with XAS_DP_OVERLAP=1 the rest runs beside backward on '
                            'the communication stream; syncbn_host_ms = host-visible time of the per-layer statistic exchanges (gloo '
                            'blocks the host; RCCL only enqueues, the stream-side cost is inside step_ms)'}
    sync()
    # one extra, UNTIMED step with the side stream disabled: the same kernels measured without concurrent
    # neighbours (kernel quality), beside the overlapped figures of the timed region (step throughput)
    from xas_amd import ops_nn as _ops
    serial = KernelTimer()
    _side_was = _ops._side['enabled']
    _ops._side['enabled'] = False
    with serial:
        step(x)
    sync()
    _ops._side['enabled'] = _side_was

    # the exact-fp32 MFMA figure beside the headline (VERDICT r02 ruling, condition d): the same step with every MFMA
    # convolution on v_mfma_f32_32x32x2_f32, after the timed region (its weight copies are rebuilt in that format)
    f32_ms = None
    if args.precision != 'f32' and args.f32_steps > 0:
        _xl.query('xas_set_precision', _xl.PREC_F32)
        step(x)                                          # warm-up: re-packs the weights for the exact-fp32 kernels
        sync()
        tf0 = time.perf_counter()
        for _ in range(args.f32_steps):
            step(x)
        sync()
        f32_ms = (time.perf_counter() - tf0) / args.f32_steps * 1e3
        _xl.query('xas_set_precision', _xl.PREC_NAMES[args.precision])
        log('exact-fp32 MFMA kernels: %.1f ms/step (%d steps, untimed by the headline)' % (f32_ms, args.f32_steps))

    # beside the headline, untimed by it: the step with the real-image detector forward computed ONCE (TrainStep(dedupe=True)).  The
    # reference runs it twice per step with identical weights (modules/model.py:231 detached for the discriminator update, :64 for the
    # generator losses); computing it once and replaying the skipped pass's running-statistic updates leaves parameters AND buffers
    # bit-identical (tests/test_gpu_model.py) - an exact common-subexpression elimination a user may switch on.  NOT the headline:
    # the timed region above runs all 12 detector forwards per sample, as the reference does.
    cse_ms = None
    if not args.dedupe and args.cse_steps > 0 and opt_disc is not None and world == 1:
        step.dedupe = True
        for _ in range(2):
            step(x)
        sync()
        tc0 = time.perf_counter()
        for _ in range(args.cse_steps):
            step(x)
        sync()
        cse_ms = (time.perf_counter() - tc0) / args.cse_steps * 1e3
        step.dedupe = False
        log('real-image detector forward computed once (dedupe, bit-identical state): %.1f ms/step (%d steps, untimed by the headline)'
            % (cse_ms, args.cse_steps))

    per_sample = IMAGES_PER_SAMPLE['MPI' if args.workload.startswith('MPI') else 'HM36']
    samples = world * args.batch * args.steps
    checksums = None
    if world > 1:
        # replicas must hold identical parameters after the averaged steps: one float64 checksum per rank and optimizer
        mine = torch.tensor([float(opt_det.param_arena.double().sum()),
                             float(opt_disc.param_arena.double().sum()) if opt_disc is not None else 0.0], device=dev, dtype=torch.float64)
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        checksums = [[float(v) for v in c.tolist()] for c in allc]
    if rank == 0:
        summ = timer.summary()
        head = {k: summ.pop(k) for k in list(summ) if k.startswith('xas_head_')}
        bnfam = {k: summ.pop(k) for k in list(summ) if k.startswith('xas_bn_')}

        def classes(sm):
            """per kernel class (':f32', ':bf16', ':bf16x6'): launches, ms, flops, achieved TFLOP/s, peak, fraction"""
            out = {}
            for k, v in sm.items():
                cls = ':' + k.rsplit(':', 1)[1]
                if cls not in PEAK_BY_CLASS:
                    continue
                d = out.setdefault(cls, {'launches': 0, 'ms': 0.0, 'flops': 0.0})
                d['launches'] += v['launches']; d['ms'] += v['ms']; d['flops'] += v['flops']
            for cls, d in out.items():
                d['achieved'] = d['flops'] / (d['ms'] * 1e-3) / 1e12 if d['ms'] > 0 else 0.0
                d['peak'] = PEAK_BY_CLASS[cls]
                d['frac'] = d['achieved'] / d['peak']
            return out

        cl = classes(summ)
        dom = max(cl, key=lambda c: cl[c]['ms'])                 # the family the step spends its conv time in
        fl = sum(d['flops'] for d in cl.values())
        ms = sum(d['ms'] for d in cl.values())
        n_launch = sum(d['launches'] for d in cl.values())
        # whole conv family against the MIX of peaks: time each class would need at its own peak / time it took
        mix_frac = sum(d['flops'] / (d['peak'] * 1e12) for d in cl.values()) / (ms * 1e-3) if ms > 0 else 0.0
        scl = classes(serial.summary())
        traffic = None
        tsrc = None
        for tname in ('r05_conv_traffic.json', 'r04_conv_traffic.json', 'r03_conv_traffic.json', 'r02_conv_traffic.json'):
            tpath = os.path.join(ROOT, 'profiles', tname)
            if os.path.exists(tpath) and args.workload == 'HM36_Multi_SurS1' and args.batch == 32:
                with open(tpath) as tf:
                    pmc = json.load(tf)
                if pmc.get('precision', 'f32') == args.precision:
                    traffic = pmc.get('bytes_per_launch')
                    tsrc = ('profiles/%s: static constant from separate rocprofv3 --pmc passes of this command (%s), NOT '
                            'measured in this run' % (tname, pmc.get('collected', 'see file')))
                break
        variant_check = None
        if args.precision != 'f32' and not args.no_variant_check:
            # this mode's distance to the exact-fp32 MFMA path, measured here (untimed): joints of one detector pass on 8 images
            reg = model.regressor
            was = reg.training
            reg.eval()                                  # running statistics: both passes see the same normalisation
            img = x['cam_%s_img' % cams[0]][:8]
            with torch.no_grad():
                kv = reg(img)[0].clone()
                _xl.query('xas_set_precision', _xl.PREC_F32)
                k32 = reg(img)[0]
                _xl.query('xas_set_precision', _xl.PREC_NAMES[args.precision])
            reg.train(was)
            variant_check = {'max_abs_joint_diff_vs_exact_fp32_mfma': float((kv - k32).abs().max()),
                             'what': 'detector forward (eval-mode norms) on 8 images, this precision mode vs the exact-fp32 MFMA kernels; parity bar 1e-4'}
            if world == 1:
                variant_check['train_step'] = train_step_variant_check(step, x, model, disc, opt_det, opt_disc, _xl, args.precision, args.tune)
        kernel_names = {':bf16x6': 'igemm_x6_kernel<.,.,.,3> (fwd / dgrad) + wgrad_x6_kernel<.,.,3>: bf16x6 MFMA implicit-GEMM conv family',
                        ':f16x3': 'igemm_x6_kernel<.,.,.,2> / igemm_x6t_kernel<.,.,2> (fwd / dgrad) + wgrad_x6_kernel<.,.,2> / wgrad_x6t_kernel<.,2>: f16x3 MFMA implicit-GEMM conv family',
                        ':bf16': 'igemm_x6_kernel<.,.,.,1> + wgrad_x6_kernel<.,.,1>: bf16 MFMA implicit-GEMM conv family',
                        ':f32': 'igemm_buf_kernel + wgrad_buf_kernel (+ stem_fwd_kernel): exact-fp32 MFMA implicit-GEMM conv family'}
        line = {
            'metric': 'images/sec %s 256px bs%d (full disc+gen training step)' % (args.workload, args.batch),
            'value': samples * per_sample / dt, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'f32': 'f32 (exact fp32 MFMA)', 'bf16x6': 'f32 (bf16x6 split, f32 accumulate)',
                      'f16x3': 'f32 (f16x3 split: every fp32 operand as two fp16 pieces, three products; f32 accumulate)',
                      'bf16': 'bf16 products, f32 accumulate (variant: NOT fp32 accurate, not the headline)'}[args.precision],
            'data': 'synthetic',
            'config': {'workload': args.workload, 'batch_per_gpu': args.batch, 'cameras': len(cams),
                       'image': '256x256', 'images_per_sample': per_sample, 'parallelism': 'dp%d' % world,
                       'samples_per_s': samples / dt, 'ranks': world,
                       **({'param_checksum_per_rank': checksums, 'replicas_identical': all(c == checksums[0] for c in checksums)}
                          if checksums else {}),
                       'samples_per_s_per_rank': samples / dt / world,
                       'detector_forwards_per_s': samples * (2 if args.dedupe else 3) * len(cams) / dt,
                       'precision': args.precision,
                       'dedupe': bool(args.dedupe), **({'variant_check': variant_check} if variant_check else {})},
            'roofline': {'bound': 'mfma', 'achieved': cl[dom]['achieved'], 'peak': cl[dom]['peak'], 'unit': 'TFLOP/s',
                         'frac': cl[dom]['frac'], 'traffic': traffic, 'traffic_source': tsrc,
                         'kernel': kernel_names[dom],
                         'what': 'dominant kernel family of the step: algorithmic FLOP (2 N Ho Wo Cout R S Cin per launch) of its '
                                 'launches in the event-timed step / the sum of their HIP-event durations on the stream each was '
                                 'launched on; peak: fp32 MFMA 157.3, bf16 / fp16 MFMA 2516.8, bf16x6 = 2516.8 / 6 = 419.5, f16x3 = 2516.8 / 3 = 838.9 TFLOP/s '
                                 'fp32-equivalent (MI355X_MICROARCH.md)',
                         'by_kernel_class': {k.lstrip(':'): {kk: d[kk] for kk in ('launches', 'ms', 'achieved', 'peak', 'frac')} for k, d in cl.items()},
                         'conv_family_frac_of_mixed_peak': mix_frac,
                         # the same rate against the peak of the round's previous fp32-accurate scheme (bf16x6: six products per fp32
                         # product, 419.5): how the fp32-equivalent throughput moved across the change of scheme, NOT the roofline fraction
                         'achieved_over_bf16x6_peak': cl[dom]['achieved'] / (PEAK_BF16_MFMA_TFLOPS / 6.0),
                         'sustained_mfma_ceiling': {
                             'bf16_TFLOPs': 1892.0, 'bf16x6_equivalent_TFLOPs': 315.3, 'f16x3_equivalent_TFLOPs': 630.7,
                             'what': 'static constant, NOT measured in this run: a register-only v_mfma_f32_32x32x16_bf16 loop '
                                     'with random (non-zero) operands sustains 1 892 TFLOP/s on this part (2 486 with zero '
                                     'operands: the clock drops under the power limit), tools/micro/mfma_bf16_peak.hip, '
                                     'profiles/r03_mfma_bf16_sustained_peak.txt (the fp16 instruction has the same rate; / 3 and / 6 for '
                                     'the split modes); frac above is against the NOMINAL peak of the class'},
                         'conv_family_tflops': fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                         'algorithmic_bytes_per_launch': timer.bytes_total / max(1, n_launch),
                         'note': 'weight-gradient kernels run on a side stream concurrently with the main chain, so '
                                 'per-launch durations include sharing; step_conv_tflops_over_wall is the whole-step view',
                         'serial': {'what': 'same launches, one extra untimed step without stream overlap (kernel quality)',
                                    'by_kernel_class': {k.lstrip(':'): {kk: d[kk] for kk in ('launches', 'ms', 'achieved', 'peak', 'frac')} for k, d in scl.items()}},
                         'step_conv_tflops_over_wall': sum(v['flops'] for v in summ.values()) / timed_steps / (dt / args.steps) / 1e12,
                         'launches_per_step': n_launch / timed_steps, 'avg_launch_us': ms * 1e3 / max(1, n_launch),
                         'event_timed_steps': timed_steps,
                         'conv_ms_per_step': ms / timed_steps,
                         'families': {k: {'launches': v['launches'] // timed_steps, 'ms_per_step': v['ms'] / timed_steps,
                                          'tflops': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] > 0 else 0.0}
                                      for k, v in summ.items()}},
        }
        if comm is not None:
            if args.ref_n1:
                comm['per_rank_over_n1'] = (samples / dt / world) / args.ref_n1
                comm['ref_n1_samples_per_s'] = args.ref_n1
            line['comm'] = comm
        if cse_ms is not None:
            line['dedupe_real_forward'] = {'ms_per_step': cse_ms, 'images_per_s': world * args.batch * per_sample / (cse_ms * 1e-3),
                                           'steps': args.cse_steps,
                                           'what': 'TrainStep(dedupe=True): the real-image detector forward that the reference runs twice per '
                                                   'step with identical weights is computed once, the skipped pass\'s running-statistic updates '
                                                   'are replayed; parameters and buffers after the step are bit-identical (tests/test_gpu_model.py). '
                                                   '8 instead of 12 detector forwards per sample: reported beside the headline, never as it'}
        if f32_ms is not None:
            line['exact_fp32_mfma'] = {'ms_per_step': f32_ms, 'images_per_s': world * args.batch * per_sample / (f32_ms * 1e-3),
                                       'steps': args.f32_steps, 'peak_TFLOPs': PEAK_FP32_MFMA_TFLOPS,
                                       'what': 'the same step with every MFMA convolution on v_mfma_f32_32x32x2_f32 '
                                               '(bench.py --precision f32 is that mode as the timed run)'}
        # soft-argmax head (HBM bound): entry-point time (partial + finalize kernels) of the calls in the event-timed step
        hd = {}
        for k, v in head.items():
            if v['ms'] > 0:
                tbs = v['flops'] / (v['ms'] * 1e-3) / 1e12
                hd[k.replace('xas_head_softargmax_', '').replace(':direct', '').replace('_amax', '').replace('from_partials', 'fwd_second_pass')] = {
                    'launches': v['launches'], 'us_per_launch': v['ms'] * 1e3 / v['launches'],
                    'algorithmic_MB_per_launch': v['flops'] / v['launches'] / 1e6, 'achieved_TBps': tbs, 'frac_of_8TBps': tbs / 8.0}
        line['roofline']['head'] = dict(hd, kernel='fwd: first pass (online softmax, marginals) in the epilogue of the final 1x1 convolution '
                                        '(igemm_x6_kernel<64,256,0,.>, xas_conv_fwd_head: the logits are NOT read again) + head_finalize_kernel '
                                        'over its records (fwd_second_pass; head_partial_kernel + head_finalize_kernel = `fwd` when the '
                                        'fused form is off or not taken); bwd: head_bwd_coef_kernel + head_bwd_kernel', bound='hbm', peak_TBps=8.0,
                                        note='logits of one grouped detector pass (18.87 MB per image; forward: every image of the pass - with the '
                                             'joint prefix pass 3 x cameras x B = %d, else 2 x cameras x B -, read once; backward: the 2 x cameras x B = %d '
                                             'graph images, read + written)' % (3 * args.batch * len(cams), 2 * args.batch * len(cams)))
        # batch-norm family (HBM bound; the second-largest family of the step): algorithmic bytes of every call of the event-timed
        # step / its HIP-event time on its stream (statistics that ride in a conv epilogue are not in here: they are conv time)
        bn_ms = sum(v['ms'] for v in bnfam.values())
        bn_by = sum(v['flops'] for v in bnfam.values())
        if bn_ms > 0:
            line['roofline']['batch_norm'] = {
                'bound': 'hbm', 'launches': sum(v['launches'] for v in bnfam.values()), 'ms_per_step': bn_ms / timed_steps,
                'algorithmic_GB_per_step': bn_by / timed_steps / 1e9, 'achieved_TBps': bn_by / (bn_ms * 1e-3) / 1e12,
                'peak_TBps': 8.0, 'frac': bn_by / (bn_ms * 1e-3) / 1e12 / 8.0,
                'by_entry': {k: {'launches': v['launches'], 'ms': v['ms'], 'TBps': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] > 0 else 0.0}
                             for k, v in bnfam.items()},
                'kernel': 'col_reduce(_lean)_kernel, bn_apply_stream_kernel, bn_bwd_apply_stream_kernel (bn.hip)',
                'note': ('per-call times include sharing the chip with the weight-gradient stream' if _ops._side['enabled'] else
                         'one stream: the calls run alone on the chip')}
        if traffic is not None:
            line['roofline']['mfma_busy_pct_pmc'] = pmc.get('mfma_busy_pct')      # same static source as `traffic`
        if args.shape_report:
            with open(args.shape_report, 'w') as f:
                f.write('entry (N,Hi,Wi,Cin,Cout,R,stride) launches ms_total TFLOP/s\n')
                for name, sig, n, ms_, tf in timer.by_shape():
                    f.write('%-16s %-36s %5d %9.3f %7.2f\n' % (name, sig, n, ms_, tf))
            with open(args.shape_report + '.serial', 'w') as f:       # the extra step without stream overlap
                f.write('entry (N,Hi,Wi,Cin,Cout,R,stride) launches ms_total TFLOP/s\n')
                for name, sig, n, ms_, tf in serial.by_shape():
                    f.write('%-16s %-36s %5d %9.3f %7.2f\n' % (name, sig, n, ms_, tf))
        if not args.no_cpu_baseline and world == 1:       # rank 0 at N = 1 only (the other ranks would sit in a barrier)
            threads = host_threads()
            log('timing the CPU oracle (B=2: 3 warm-up + 5 timed steps; B=8: 1 + 3) on %d host threads' % threads)
            line['cpu_baseline'] = cpu_baseline(args.workload, threads, B=2, warm=3, timed=5)
            line['cpu_baseline_b8'] = cpu_baseline(args.workload, threads, B=8, warm=1, timed=3)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
