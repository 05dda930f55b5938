#!/usr/bin/env python3
"""Evaluation throughput on one MI355X: detector in eval mode on every camera + the device evaluation path
(selection, triangulation, metrics), synthetic consistent scene.  usage: python tools/bench_eval.py [B] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import torch

import eval as xeval
from xas_amd import engine
from xas_amd.synthetic import model_config, synthetic_eval_batch


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cfg = model_config('HM36_Multi_SurS1')
    cfg['dataset_params'] = {'dataset': {'name': 'h36m'}, 'cam_id_list': cfg['model_params']['cam_id_list']}
    model, *_ = engine.prepare_model(cfg)
    dev = torch.device('cuda')
    ev = xeval.Eval(cfg, model.regressor, [], '/tmp')
    cams = cfg['model_params']['cam_id_list']
    x = synthetic_eval_batch(B, cams, dev, seed=1)
    kps = {}
    with torch.no_grad():
        for _ in range(2):
            out = ev.eval_batch(x, 'best')
        for c in cams:
            kps['cam_%d' % c] = ev.detector(x['cam_%d_img' % c])[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = ev.eval_batch(x, 'best')
        host = {k: v.cpu() for k, v in out.items() if not k.startswith('world')}
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(20):
        out = ev.eval_batch(x, 'best', kps_by_cam=kps)
    torch.cuda.synchronize()
    dp = (time.perf_counter() - t0) / 20
    print('eval batch B=%d x %d cameras: %.1f ms (%.0f images/s); evaluation path alone (select + patch->world + '
          'triangulation + metrics): %.3f ms' % (B, len(cams), dt * 1e3, B * len(cams) / dt, dp * 1e3))


if __name__ == '__main__':
    main()
