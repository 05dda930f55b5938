"""Repeat the two-rank data-parallel step (tests/test_gpu_dp_step.py) N times per option set and log every outcome: a stuck
rank dumps the Python stacks of all its threads (tests/_ranks.py) into the log.   python tools/loop_dp_step.py N [limit_s]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import _ranks                      # noqa: E402
import test_gpu_dp_step as t       # noqa: E402

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    cases = [dict(beside=1), dict(beside=0), dict(beside=1, cams=[0, 1], dedupe=False), dict(beside=1, notify=0)]
    sel = os.environ.get('LOOP_CASES')
    if sel:
        cases = [cases[int(i)] for i in sel.split(',')]
    bad = 0
    for i in range(n):
        for c in cases:
            t0 = time.time()
            try:
                r = _ranks.run_ranks(t._worker, 2, (c,), limit=limit)
                ok = r[0][:3] == r[1][:3] and r[0][5]
                print('iter %d %s: %s in %.1f s' % (i, c, 'ok' if ok else 'MISMATCH %s' % (r,), time.time() - t0), flush=True)
                bad += 0 if ok else 1
            except AssertionError as e:
                bad += 1
                print('iter %d %s: FAILED after %.1f s\n%s' % (i, c, time.time() - t0, e), flush=True)
    print('done: %d bad' % bad, flush=True)
    sys.exit(1 if bad else 0)
