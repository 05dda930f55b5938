"""HIP stream plumbing for the multi-camera step.

The per-camera pipelines of one step (detector -> geometry -> mask renderer -> physique net -> mask losses, and
the pseudo-image branch) are independent until the losses are summed, and most of their kernels are too small
to fill 256 CUs.  Each camera therefore runs on its own HIP stream (forward AND backward: autograd replays a
node on the stream it was recorded on), next to the weight-gradient side stream of ops_nn.

Order-dependent state is kept deterministic:
  * batch-norm running statistics and `num_batches_tracked` are updated on ONE bookkeeping stream in host
    program order (camera 0 first), exactly the order of the single-stream reference;
  * packed weight copies are (re)built on the main stream before the cameras fork.
The fan-out of the passes that are followed by a backward is OFF by default (XAS_CAM_STREAMS=1): at B=32 the kernels
of one camera already fill the chip and in the backward the extra streams meet the weight-gradient stream
(tools/ab_step.py: 1 stream 258 ms/step, 4 streams 259); the weight-gradient side stream of ops_nn is what pays
(327 -> 306 ms at the time).  The gradient-free detector passes of the discriminator step can fan out separately
(XAS_NOGRAD_STREAMS=4): batch-norm kernels of one camera fill the matrix-pipe gaps of another, -0.6 % step time in
bench.py (-1.4 % in tools/ab_step.py); off by default because the overlapped launches read 4 % slower per launch in
the roofline measurement for a gain inside the box-to-box noise, and never used in data-parallel runs (SyncBatchNorm
collectives stay on one stream).
"""
import os

import torch
import torch.distributed

NUM = max(1, int(os.environ.get('XAS_CAM_STREAMS', '1')))   # measured on MI355X: 1 -> 306 ms/step, 2 -> 312, 4 -> 321
# Fan-out of the gradient-free detector passes of the discriminator step (no backward follows, so the extra streams do
# not meet the weight-gradient stream): tools/ab_step.py on MI355X: 1 -> 257.5 ms/step, 2 -> 259.0, 3 -> 253.8, 4 -> 253.9;
# bench.py (10 steps): 1 -> 257.3 / 260.0, 4 -> 255.8 / 258.3.  Default 1 (see the module docstring).
NUM_NOGRAD = max(1, int(os.environ.get('XAS_NOGRAD_STREAMS', '1')))
_num = [NUM]
_cam = []
_book = [None]
_active = [False]          # True while camera streams are forked (BN bookkeeping must use the book stream)
_main = [None]             # the stream that forked (consumer of what the camera streams produce)


def enabled():
    return _num[0] > 1 and torch.cuda.is_available()


def cam_stream(i):
    while len(_cam) < _num[0]:
        _cam.append(torch.cuda.Stream())
    return _cam[i % _num[0]]


def book_stream():
    if _book[0] is None:
        _book[0] = torch.cuda.Stream()
    return _book[0]


def forked():
    return _active[0]


class fork:
    """Context: camera streams may be used inside; on exit the main stream waits for all of them and for the
    bookkeeping stream."""

    def __init__(self, num=None):
        self.num = NUM if num is None else num
        if num is not None and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            self.num = NUM      # data-parallel runs keep the SyncBatchNorm collectives on one stream

    def __enter__(self):
        self.main = torch.cuda.current_stream()
        self.prev, _num[0] = _num[0], self.num
        _active[0] = enabled()
        _main[0] = self.main
        if _active[0]:
            book_stream().wait_stream(self.main)
        return self

    def run(self, i):
        """Stream context for camera i (the main stream when the fan-out is off)."""
        if not _active[0]:
            return torch.cuda.stream(self.main)
        s = cam_stream(i)
        s.wait_stream(self.main)
        return torch.cuda.stream(s)

    def __exit__(self, *exc):
        if _active[0]:
            for s in _cam:
                self.main.wait_stream(s)
            self.main.wait_stream(book_stream())
        _active[0] = False
        _num[0] = self.prev
        _main[0] = None


def to_main(*tensors):
    """Tell the caching allocator that tensors produced on a camera stream are consumed on the main stream."""
    if not enabled():
        return
    main = _main[0] if _main[0] is not None else torch.cuda.current_stream()
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(main)


# ---- pass-level chains (round 3) -------------------------------------------------------------------------------------
# With the bf16-split convolution kernels the step is no longer bound by the matrix pipe alone: on ONE stream the chain of a
# pass is conv (matrix pipe) -> batch norm (HBM) -> conv -> ..., each kernel waiting for the one before, so the matrix pipe
# idles during the norms and HBM during the convolutions (r03 timeline: main stream 156 ms busy = 87 conv + 48 norm + 21
# other).  The generator step has two INDEPENDENT detector passes - the real images of all cameras (followed by geometry,
# mask renderer, physique net) and the pseudo images - and autograd replays a node on the stream it was recorded on: each
# runs as its own chain on its own stream, forward and backward, and the norms of one chain fill the matrix-pipe time of
# the other.  Order-dependent state stays deterministic exactly as for camera streams: running statistics and batch
# counters are updated on the bookkeeping stream in host program order (real pass first, as the reference's call order);
# weight gradients of both chains go to the ONE side stream in host order; parameter gradients of the norms are added with
# hardware atomics (one contribution per chain from a zeroed arena: order-independent bit for bit).
# MEASURED (r03, MI355X, HM36_Multi_SurS1 B = 32): XAS_CHAINS=2 166.0 / 166.3 ms per step against 159.9 on one stream.  The
# kernels of two chains do run side by side (per-launch durations of the forward convolutions grow from 23.9 to 40.4 ms per
# step in total), but the sum does not shrink: at bf16x6 speed the convolution kernels are themselves within reach of the
# HBM bound (~500 GB of traffic per step = ~100 ms at 5 TB/s against ~90 ms of matrix-pipe time), so a norm running beside a
# convolution takes its bandwidth rather than idle time.  Default 1; the mode stays tested (tests/test_gpu_groups.py).
CHAINS = max(1, int(os.environ.get('XAS_CHAINS', '1')))
_chain = []
_chain_used = []


def chain_stream(i):
    while len(_chain) <= i:
        _chain.append(torch.cuda.Stream())
    return _chain[i]


def chain_streams_in_use():
    """Streams that carried backward-relevant work of the current step (the data-parallel reducer waits on all of them)."""
    return list(_chain_used)


def reset_chain_use():
    del _chain_used[:]


class chains:
    """Context: `run(i)` gives the stream context of chain i (the current stream when chains are off); on exit the
    entering stream waits for every chain and for the bookkeeping stream."""

    def __init__(self, num=None):
        self.num = CHAINS if num is None else num
        if not torch.cuda.is_available():
            self.num = 1

    def __enter__(self):
        self.main = torch.cuda.current_stream()
        self.was, self.was_main = _active[0], _main[0]
        self.used = []
        if self.num > 1:
            _active[0] = True
            _main[0] = self.main
            book_stream().wait_stream(self.main)
        return self

    def run(self, i):
        if self.num <= 1:
            return torch.cuda.stream(self.main)
        s = chain_stream(i % self.num)
        s.wait_stream(self.main)
        if s not in self.used:
            self.used.append(s)
        if s not in _chain_used:
            _chain_used.append(s)
        return torch.cuda.stream(s)

    def to_main(self, *tensors):
        """Tensors produced on a chain and consumed on the entering stream: tell the caching allocator."""
        if self.num <= 1:
            return
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(self.main)

    def __exit__(self, *exc):
        if self.num > 1:
            for s in self.used:
                self.main.wait_stream(s)
            self.main.wait_stream(book_stream())
            _active[0] = self.was
            _main[0] = self.was_main              # (nested inside an active camera-stream fork: the outer context keeps its main stream)
