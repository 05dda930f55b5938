"""The optimisation step as a function of its inputs and nothing else (VERDICT r04 "next" 1a).

* run twice from the same state in one process, and once in a second process: gradient arenas as Adam consumes them,
  losses, running statistics, parameters after the update and the int64 depth-peak indices are BIT-identical;
* the same with every free block of the caching allocator filled with 1e30 / NaN / -1 before the step: a kernel that reads
  memory nobody wrote (a stale workspace, an unwritten maximum, a buffer used before its producer ran) shows up as a
  difference - 1e30 enters a recorded maximum (fmaxf drops NaN), NaN enters every sum.
Both at the BASELINE size (B = 32, HM36 S1 and S2); the poison legs loop, because a read that races with its producer
only sometimes loses.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

import _stepcheck as sc

pytestmark = [pytest.mark.gpu, pytest.mark.selfcheck]
HERE = os.path.dirname(os.path.abspath(__file__))


def _same(a, b, what):
    for k in sc.TENSOR_KEYS:
        if k in a:
            ne = int((a[k] != b[k]).sum()) if a[k].dtype.is_floating_point else int((a[k] != b[k]).sum())
            # (NaN != NaN: a NaN anywhere counts as a difference, as it should)
            assert ne == 0, '%s: %s differs in %d of %d elements' % (what, k, ne, a[k].numel())


@pytest.mark.limit(400)
@pytest.mark.parametrize('name', ['HM36_Multi_SurS1', 'HM36_Multi_SurS2'])
def test_step_is_bitwise_reproducible(name):
    from xas_amd import state
    step, x = sc.build_step(name, 32)
    first = sc.run_captured(step, x)                    # from the freshly built state: what a second process computes too
    assert bool(torch.isfinite(first['det']).all()) and bool(torch.isfinite(first['loss']).all())
    assert first['peaks'].numel() == 3 * 4 * 32 * 18 * 3          # every detector call of the step reported its indices
    # second process, same construction: digests of the same quantities
    from _ranks import release_gpu_memory
    release_gpu_memory()                     # the child needs the card's memory, not this process's cache
    p = subprocess.run([sys.executable, os.path.join(HERE, '_stepcheck.py'), name, '32'], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('STEPCHECK ')][-1]
    other = json.loads(line[len('STEPCHECK '):])
    mine = sc.digest(first)
    assert other == mine, {k: (other[k][:12], mine[k][:12]) for k in mine if other.get(k) != mine[k]}
    # same process, from a snapshot taken after that step (moments, running statistics no longer at their initial values)
    sn = state.snapshot(step)
    a = sc.run_captured(step, x, sn)
    for i in range(3):
        _same(sc.run_captured(step, x, sn), a, '%s run %d vs run 0' % (name, i + 1))


@pytest.mark.limit(600)
@pytest.mark.parametrize('name', ['HM36_Multi_SurS1', 'HM36_Multi_SurS2'])
def test_step_reads_no_memory_it_did_not_write(name):
    from xas_amd import state
    step, x = sc.build_step(name, 32)
    sc.run_captured(step, x)
    sn = state.snapshot(step)
    ref = sc.run_captured(step, x, sn)
    for i, value in enumerate([1e30, float('nan'), -1.0] + [1e30] * 9):
        nbytes = sc.poison_free_memory(value)
        assert nbytes > 8 << 30                          # the step's activations went back to the cache: tens of GB to fill
        _same(sc.run_captured(step, x, sn), ref, '%s, free memory filled with %r (leg %d)' % (name, value, i))
