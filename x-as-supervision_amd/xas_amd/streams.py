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
