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

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, 'Peak FP32 (matrix)'
IMAGES_PER_SAMPLE = {'HM36': 8, 'MPI': 10}


def host_threads():
    """Cores this process may use: the GPU box gives one GPU's share (16), not the host's core count."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(workload, threads):
    """Oracle (CPU restatement of the reference step, stock PyTorch fp32) on a bounded sample: ONE full
    disc+gen step at B=8 on the host cores."""
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
    B = 8                      # ~10 s of host work on a 16-core share; the GPU leg runs B = 32
    x = synthetic_batch(B, cfg['cam_id_list'], torch.device('cpu'), seed=1)
    t0 = time.perf_counter()
    ostep.train_step(cfg, reg, phys, disc, o_det, o_disc, x)
    dt = time.perf_counter() - t0
    per_sample = IMAGES_PER_SAMPLE['MPI' if workload.startswith('MPI') else 'HM36']
    return {'value': B * per_sample / dt, 'unit': 'images/s', 'cores': threads, 'kind': 'port',
            'sample': '1 full disc+gen step incl. Adam, %s, B=%d, fp32, oracle on torch CPU (%.1f s)' % (workload, B, dt)}


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
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16', 'bf16x6'],
                    help='f32 (default, the headline: exact fp32 MFMA) or bf16: NOT the headline - forward / data-gradient '
                         'convolutions on bf16 MFMA (fp32 accumulate, fp32 master weights, fp32 weight gradients); the JSON line '
                         'carries dtype "bf16" and peak = the bf16 MFMA peak for those launches')
    ap.add_argument('--dedupe', action='store_true',
                    help='NOT the headline: share the real-image detector forward between the discriminator and the '
                         'generator update (engine.TrainStep(dedupe=True)); the JSON line is marked config.dedupe')
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
        if args.backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    from xas_amd import engine
    from xas_amd.prof import KernelTimer
    from xas_amd.synthetic import model_config, synthetic_batch
    from xas_amd import _lib as _xl
    _xl.query('xas_set_tuning', args.tune)
    _xl.query('xas_set_precision', {'f32': 0, 'bf16': 1, 'bf16x6': 2}[args.precision])
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
    HEAD = ('xas_head_softargmax_fwd', 'xas_head_softargmax_bwd')
    timer = KernelTimer(CONV_ENTRIES + HEAD)
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

    # one extra, UNTIMED step with the side stream disabled: the same kernels measured without concurrent
    # neighbours (kernel quality), beside the overlapped figures of the timed region (step throughput)
    from xas_amd import ops_nn as _ops
    serial = KernelTimer()
    _ops._side['enabled'] = False
    with serial:
        step(x)
    sync()
    _ops._side['enabled'] = True

    per_sample = IMAGES_PER_SAMPLE['MPI' if args.workload.startswith('MPI') else 'HM36']
    samples = world * args.batch * args.steps
    if rank == 0:
        summ = timer.summary()
        head = {k: summ.pop(k) for k in list(summ) if k.startswith('xas_head_')}
        mfma = {k: v for k, v in summ.items() if not k.endswith(':direct')}
        fl = sum(v['flops'] for v in mfma.values())
        ms = sum(v['ms'] for v in mfma.values())
        n_launch = sum(v['launches'] for v in mfma.values())
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'r02_conv_traffic.json')
        if os.path.exists(tpath) and args.workload == 'HM36_Multi_SurS1' and args.batch == 32:
            with open(tpath) as tf:
                pmc = json.load(tf)
                traffic = pmc.get('bytes_per_launch')   # PMC passes of this same command (see file)
        variant_check = None
        if args.precision != 'f32':
            # the variant's distance to the fp32-MFMA path, measured here (untimed): joints of one detector pass on 8 images
            reg = model.regressor
            was = reg.training
            reg.eval()                                  # running statistics: both passes see the same normalisation
            img = x['cam_%s_img' % cams[0]][:8]
            with torch.no_grad():
                kv = reg(img)[0].clone()
                _xl.query('xas_set_precision', 0)
                k32 = reg(img)[0]
                _xl.query('xas_set_precision', {'bf16': 1, 'bf16x6': 2}[args.precision])
            reg.train(was)
            variant_check = {'max_abs_joint_diff_vs_f32_path': float((kv - k32).abs().max()),
                             'what': 'detector forward (eval-mode norms) on 8 images, this precision mode vs the fp32-MFMA path; parity bar 1e-4'}
        line = {
            'metric': 'images/sec %s 256px bs%d (full disc+gen training step)' % (args.workload, args.batch),
            'value': samples * per_sample / dt, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'f32': 'f32', 'bf16': 'bf16 fwd/dgrad MFMA + f32 wgrad (variant, not the headline)',
                      'bf16x6': 'f32 products from 6 bf16 MFMA partial products in fwd/dgrad + f32 wgrad (variant, not the headline)'}[args.precision],
            'data': 'synthetic',
            'config': {'workload': args.workload, 'batch_per_gpu': args.batch, 'cameras': len(cams),
                       'image': '256x256', 'images_per_sample': per_sample, 'parallelism': 'dp%d' % world,
                       'samples_per_s': samples / dt,
                       'detector_forwards_per_s': samples * (2 if args.dedupe else 3) * len(cams) / dt,
                       'dedupe': bool(args.dedupe), **({'variant_check': variant_check} if variant_check else {})},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_FP32_MFMA_TFLOPS, 'traffic': traffic,
                         'algorithmic_bytes_per_launch': timer.bytes_total / max(1, n_launch),
                         'kernel': 'igemm_kernel / wgrad_kernel (fp32 MFMA implicit-GEMM conv family)',
                         'note': 'weight-gradient kernels run on a side stream concurrently with the main chain, so '
                                 'per-launch durations include sharing; step_conv_tflops_over_wall is the whole-step view',
                         'serial': (lambda sm: {'achieved': sum(v['flops'] for k, v in sm.items() if not k.endswith(':direct')) /
                                                (sum(v['ms'] for k, v in sm.items() if not k.endswith(':direct')) * 1e-3) / 1e12,
                                                'unit': 'TFLOP/s', 'what': 'same launches, one extra untimed step without stream overlap'})(serial.summary()),
                         'step_conv_tflops_over_wall': sum(v['flops'] for v in summ.values()) / timed_steps / (dt / args.steps) / 1e12,
                         'launches_per_step': n_launch / timed_steps, 'avg_launch_us': ms * 1e3 / max(1, n_launch),
                         'event_timed_steps': timed_steps,
                         'conv_ms_per_step': ms / timed_steps,
                         'families': {k: {'launches': v['launches'] // timed_steps, 'ms_per_step': v['ms'] / timed_steps,
                                          'tflops': (v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] > 0 else 0.0}
                                      for k, v in summ.items()}},
        }
        # soft-argmax head (HBM bound): entry-point time (partial + finalize kernels) of the calls in the event-timed step
        hd = {}
        for k, v in head.items():
            if v['ms'] > 0:
                tbs = v['flops'] / (v['ms'] * 1e-3) / 1e12
                hd[k.replace('xas_head_softargmax_', '').replace(':direct', '')] = {
                    'launches': v['launches'], 'us_per_launch': v['ms'] * 1e3 / v['launches'],
                    'algorithmic_MB_per_launch': v['flops'] / v['launches'] / 1e6, 'achieved_TBps': tbs, 'frac_of_8TBps': tbs / 8.0}
        line['roofline']['head'] = dict(hd, kernel='head_partial_kernel + head_finalize_kernel (fwd), head_bwd_coef_kernel + '
                                        'head_bwd_kernel (bwd)', bound='hbm', peak_TBps=8.0,
                                        note='logits of one camera-batched pass (%d images x 18.87 MB): read once forward, read + '
                                             'written backward' % (args.batch * len(cams)))
        if traffic is not None:
            line['roofline']['mfma_busy_pct_pmc'] = pmc.get('mfma_busy_pct')
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
            log('timing the CPU oracle step (B=8) on %d host threads' % threads)
            line['cpu_baseline'] = cpu_baseline(args.workload, threads)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
