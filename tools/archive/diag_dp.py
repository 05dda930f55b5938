#!/usr/bin/env python3
"""Two gloo ranks on one GPU: after the generator backward + finish(), compare the averaged gradient arena across ranks
and name the parameters that differ (diagnostic for the early-bucket path of dp.GradReducer)."""
import os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from xas_amd import engine, ops_nn
    from xas_amd.synthetic import model_config, synthetic_batch
    cfg = model_config('HM36_Multi_SurS2')
    cfg['model_params']['cam_id_list'] = [0]
    torch.manual_seed(100 + rank)
    model, disc, od, odisc = engine.prepare_model(cfg)
    model.cuda().train(), disc.cuda().train()
    disc.smpl_discriminator.header.p = 0.0
    step = engine.TrainStep(cfg, model, disc, od, odisc, num_buckets=3)
    x = synthetic_batch(2, [0], torch.device('cuda'), seed=10 + rank)
    launched = []
    reports = {}
    names_by_id = {id(p): n for n, p in list(model.regressor.named_parameters()) + [('phys.' + n, p) for n, p in model.physique_network.named_parameters()]}
    orig_notify = step.red_det.notify
    orig_ready = step.red_det._member_ready
    seq = []
    def spy_ready(j):
        seq.append(j)
        return orig_ready(j)
    step.red_det._member_ready = spy_ready
    nseq = []
    def spy_notify(p):
        nseq.append(step.red_det._index.get(p.data_ptr()))
        return orig_notify(p)
    step.red_det.notify = spy_notify
    orig = step.red_det._launch
    def spy(b):
        launched.append((b['lo'], b['hi'], step.red_det._armed))
        return orig(b)
    step.red_det._launch = spy
    # replicate TrainStep.__call__'s generator half up to finish()
    for it in range(3):
        loss_kp, info = model(x, disc.smpl_discriminator)
        total = sum(v.mean() for v in loss_kp.values())
        launched.clear()
        seq.clear(); nseq.clear()
        pend = dict(ops_nn._uses['pending'])
        step.red_det.arm()
        total.backward()
        ops_nn.join_side_stream()
        step.red_det.finish()
        torch.cuda.synchronize()
        g = od.grad_arena.clone()
        gl = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(gl, g)
        if rank == 0:
            from collections import Counter
            c = Counter(seq)
            multi = {j: n for j, n in c.items() if n != 1}
            f = od._flat
            print('   members reported:', len(c), 'of', sum(len(b['members']) for b in step.red_det.buckets), ' reported != once:', len(multi),
                  ' pending counts seen before backward:', Counter(pend.values()), flush=True)
            # position in the report sequence of the first parameter (conv1.weight, index 0) and bucket completion points
            cn = Counter(nseq)
            print('   notify calls:', len(nseq), 'distinct', len(cn), 'None:', cn.get(None, 0), ' notify>1:', sum(1 for v in cn.values() if v > 1),
                  ' hook-only reports:', len(seq) - len(nseq), flush=True)
            if 0 in c:
                print('   conv1.weight reported at position', seq.index(0), 'of', len(seq), flush=True)
            diff = (gl[0] - gl[1]).abs()
            bad = diff > 0
            print('it', it, 'launch order (lo, hi, early):', launched, ' mismatching elements:', int(bad.sum()), flush=True)
            if bad.any():
                f = od._flat
                names = [n for n, _ in model.regressor.named_parameters()] + ['phys.' + n for n, _ in model.physique_network.named_parameters()]
                for (p, o, n) in zip(f['params'], f['offs'], names):
                    nb = int(bad[o:o + p.numel()].sum())
                    if nb:
                        print('   ', n, o, nb, 'of', p.numel(), float(diff[o:o + p.numel()].max()), flush=True)
        od.zero_grad()
    dist.destroy_process_group()


if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    ps = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    [p.start() for p in ps]
    [p.join() for p in ps]
