"""Snapshot / restore of everything one optimisation step reads and writes (engine.TrainStep): parameters, Adam moments,
gradient arenas, module buffers (batch-norm running statistics, counters), optimizer step counts, the step counter.
Measurement and test plumbing (bench.py's variant check, the reproducibility tests): the same step can be run again from
the SAME state, in another arithmetic or another schedule.
"""
import torch


def _flats(step):
    return [o._flat for o in (step.opt_det, step.opt_disc) if o is not None]


def _buffers(step):
    return [b for m in (step.model, step.disc) for b in m.buffers()]


def snapshot(step):
    # (the gradient arenas too: the generator's backward leaves gradients in the DISCRIMINATOR's arena that the next
    # discriminator update consumes - train.py:160-190 never zeroes them in between, and neither does the mirror)
    torch.cuda.synchronize()
    return ([f[k].clone() for f in _flats(step) for k in ('p', 'm', 'v', 'g')],
            [b.clone() for b in _buffers(step)],
            [o._steps for o in (step.opt_det, step.opt_disc) if o is not None], step.cur_step)


def restore(step, sn):
    ts, bs, steps, cur = sn
    it = iter(ts)
    with torch.no_grad():
        for f in _flats(step):
            for k in ('p', 'm', 'v', 'g'):
                f[k].copy_(next(it))
        for b, v in zip(_buffers(step), bs):
            b.copy_(v)
    for o, s in zip([o for o in (step.opt_det, step.opt_disc) if o is not None], steps):
        o._steps = s
        o._epoch[0] += 1                         # packed weight copies are stale
    step.cur_step = cur
    torch.cuda.synchronize()
