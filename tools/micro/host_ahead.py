#!/usr/bin/env python3
"""Is the step limited by the host?  Times the Python side of TrainStep.__call__ (the call returns when everything is ENQUEUED)
against the GPU side, over consecutive steps without any synchronisation in between, and asks after each call whether the
PREVIOUS step has already finished on the GPU (then the GPU was waiting for the host).
usage: python tools/micro/host_ahead.py [workload] [batch] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import _stepcheck  # noqa: E402  (puts the package on sys.path)
import torch  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'HM36_Multi_SurS1'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    step, x = _stepcheck.build_step(name, batch, planted=False)
    for _ in range(3):
        step(x)
    torch.cuda.synchronize()
    host, ends, done_before_next = [], [], []
    t_all = time.perf_counter()
    for i in range(steps):
        t0 = time.perf_counter()
        step(x)
        host.append((time.perf_counter() - t0) * 1e3)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        if ends:
            done_before_next.append(ends[-1].query())       # previous step finished before THIS one was fully enqueued
        ends.append(e)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) * 1e3 / steps
    gpu = [ends[i].elapsed_time(ends[i + 1]) for i in range(steps - 1)]
    print('workload %s B=%d: wall %.1f ms/step; host side of the call %.1f ms (min %.1f max %.1f); GPU step-end to step-end %.1f ms'
          % (name, batch, wall, sum(host) / steps, min(host), max(host), sum(gpu) / len(gpu)))
    print('previous step already finished when the next call returned: %d of %d' % (sum(done_before_next), len(done_before_next)))
    # where the host time goes: the same call under cProfile (one step, synchronised before and after)
    if os.environ.get('XAS_HOST_PROFILE', '1') == '1':
        import cProfile
        import pstats
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        step(x)
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        st.sort_stats('cumulative').print_stats(45)
        st.sort_stats('tottime').print_stats(30)


if __name__ == '__main__':
    main()
