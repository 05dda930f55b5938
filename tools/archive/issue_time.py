"""How far ahead of the GPU does the host run?  Times the return of every `step(x)` call (no synchronisation in between)
against the GPU time of the same steps.  If the host issues a step in much less than the GPU needs for it, launch-bound
stretches seen under rocprofv3 (whose per-launch host cost is 2-3 x) are artefacts of the profiler; if not, they are real.

    python tools/issue_time.py [--steps 8] [--batch 32] [--workload HM36_Multi_SurS1]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'x-as-supervision_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--workload', default='HM36_Multi_SurS1')
    args = ap.parse_args()
    from xas_amd import engine
    from xas_amd.synthetic import model_config, synthetic_batch
    dev = torch.device('cuda', 0)
    cfg = model_config(args.workload)
    torch.manual_seed(1234)
    model, disc, opt_det, opt_disc = engine.prepare_model(cfg)
    model.to(dev).train()
    disc.to(dev).train()
    step = engine.TrainStep(cfg, model, disc, opt_det, opt_disc)
    x = synthetic_batch(args.batch, cfg['model_params']['cam_id_list'], dev, seed=100)
    for _ in range(2):
        step(x)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    issue = []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        ts = time.perf_counter()
        step(x)
        marks[i + 1].record()
        issue.append((time.perf_counter() - ts) * 1e3)
    t_issue = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) * 1e3
    gpu = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    print('host issue per step (ms):', ' '.join('%.1f' % v for v in issue))
    print('GPU per step, main stream events (ms):', ' '.join('%.1f' % v for v in gpu))
    print('all steps issued after %.1f ms, GPU done after %.1f ms: host is %s' %
          (t_issue, t_all, 'AHEAD (GPU-bound)' if t_issue < 0.9 * t_all else 'NOT ahead (launch-bound)'))


if __name__ == '__main__':
    main()
