"""Harness for the multi-process tests: spawn `world` ranks, bound their life, never leave one behind.

A stuck rank must cost one test, not the run (r03: one hung collective held the driver's whole GPU tier):
  * every worker arms `faulthandler.dump_traceback_later` BEFORE anything can block, so a rank that does not finish
    writes the Python stack of every thread to its log and exits by itself;
  * every collective carries a timeout (`init_group`: `init_process_group(timeout=...)`, inherited by `new_group`), so a
    peer that died or a mismatched collective raises on the surviving rank instead of waiting gloo's default 30 minutes;
  * the parent joins against ONE deadline for all ranks, then terminates / kills whatever is left in a `finally`, and the
    failure message names the rank that was stuck and carries the tail of each rank's log (stderr + stack dump).
"""
import datetime
import faulthandler
import os
import socket
import sys
import tempfile
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATHS = (ROOT, os.path.join(ROOT, 'x-as-supervision_amd'), os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests'))

RANK_LIMIT_S = 150          # a rank that has not finished by then dumps its stacks and exits
COLLECTIVE_TIMEOUT_S = 90   # a collective whose peer never arrives raises after this


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def init_group(backend, rank, world, **kw):
    import torch.distributed as dist
    dist.init_process_group(backend, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S), **kw)


def _entry(target, rank, world, port, ret, args, log_path, limit):
    log = open(log_path, 'w', buffering=1)
    os.dup2(log.fileno(), 2)                    # stderr of this rank (C++ warnings, tracebacks, the stack dump)
    sys.stderr = log
    faulthandler.enable(file=log)
    faulthandler.dump_traceback_later(limit, exit=True, file=log)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for p in PATHS:
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        ret[rank] = target(rank, world, *args)
    except BaseException:
        traceback.print_exc(file=log)
        log.flush()
        os._exit(1)                              # do not wait for interpreter teardown with a half-dead process group
    faulthandler.cancel_dump_traceback_later()
    log.write('rank %d: target returned\n' % rank)      # a rank found alive after this line is stuck in interpreter / runtime teardown
    log.flush()


def release_gpu_memory():
    """Give the memory this process only CACHES back to the driver before another process is started on the same card.  After
    the full-size tests the caching allocator of the pytest process holds most of the 288 GB (r05: 287.9 GB reserved when the
    first two-rank test started, whose ranks then died of `HIP out of memory` with 0.6 GB of their own) - a child process
    cannot use what the parent merely keeps for re-use."""
    import gc
    import torch
    gc.collect()
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


class RanksStuck(AssertionError):
    """A rank did not finish inside its limit (its log holds the stack dump)."""


def run_ranks(target, world=2, args=(), limit=RANK_LIMIT_S, retry_stuck=0):
    """Run `target(rank, world, *args)` in `world` spawned processes -> {rank: return value}.  Raises AssertionError naming
    the stuck / failed ranks with their logs; no worker survives this call.  The per-rank logs of a failed run are KEPT
    (their directory is named in the message).
    retry_stuck (default 0: a hang is a FAILURE - r04 retried once and only warned, which let an unexplained hang pass): a
    run in which a rank was STUCK (stack dump / still alive at the deadline - not one that raised) is repeated that many
    times; a test that asks for it must carry `limit(>= (retry_stuck + 1) * (limit + 60))`."""
    release_gpu_memory()
    for attempt in range(retry_stuck + 1):
        try:
            return _run_ranks_once(target, world, args, limit)
        except RanksStuck as e:
            if attempt == retry_stuck:
                raise
            import warnings
            print('[tests/_ranks.py] attempt %d: %s' % (attempt + 1, e), file=sys.stderr, flush=True)
            warnings.warn('a rank of %s was stuck for %d s (stack dump in the captured stderr); retrying'
                          % (getattr(target, '__name__', target), limit))


def _run_ranks_once(target, world, args, limit):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    mgr = ctx.Manager()
    tmp = tempfile.mkdtemp(prefix='xas_ranks_')
    logs = [os.path.join(tmp, 'rank%d.log' % r) for r in range(world)]
    procs = []
    failed = False
    try:
        ret = mgr.dict()
        port = free_port()
        procs = [ctx.Process(target=_entry, args=(target, r, world, port, ret, args, logs[r], limit), daemon=True)
                 for r in range(world)]
        for p in procs:
            p.start()
        deadline = time.time() + limit + 60      # the workers' own dump fires first (the margin covers a cold `import torch`)
        for p in procs:
            p.join(max(0.1, deadline - time.time()))
        bad = [(r, p.exitcode) for r, p in enumerate(procs) if p.exitcode != 0]
        if bad:
            tails, texts = [], []
            for r in range(world):
                try:
                    with open(logs[r]) as f:
                        texts.append(f.read())
                    tails.append('---- rank %d log ----\n%s' % (r, texts[-1][-6000:]))
                except OSError:
                    pass
            # stuck = a rank still alive at the deadline, or one that left through its faulthandler dump ("Timeout (h:mm:ss)!")
            # while NO rank raised: a peer that dies of an exception also leaves the others waiting - that is a failure
            raised = any('Traceback (most recent call last)' in t for t in texts)
            stuck = (any(code is None for _, code in bad) or any('\nTimeout (' in '\n' + t for t in texts)) and not raised
            failed = True
            msg = ('ranks failed or stuck (rank, exitcode; None = still running after %d s): %s\nper-rank logs kept in %s\n%s'
                   % (limit + 60, bad, tmp, '\n'.join(tails)))
            raise (RanksStuck if stuck else AssertionError)(msg)
        return dict(ret)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(5)
            if p.is_alive():
                p.kill()
                p.join(5)
        mgr.shutdown()
        if not failed:
            for f in logs:
                try:
                    os.remove(f)
                except OSError:
                    pass
            try:
                os.rmdir(tmp)
            except OSError:
                pass
