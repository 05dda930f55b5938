"""HIP stream plumbing for the multi-camera step.

The per-camera pipelines of one step (detector -> geometry -> mask renderer -> physique net -> mask losses, and
the pseudo-image branch) are independent until the losses are summed, and most of their kernels are too small
to fill 256 CUs.  Each camera therefore runs on its own HIP stream (forward AND backward: autograd replays a
node on the stream it was recorded on), next to the weight-gradient side stream of ops_nn.

Order-dependent state is kept deterministic:
  * batch-norm running statistics and `num_batches_tracked` are updated on ONE bookkeeping stream in host
    program order (camera 0 first), exactly the order of the single-stream reference;
  * packed weight copies are (re)built on the main stream before the cameras fork.
The fan-out is OFF by default (XAS_CAM_STREAMS=1): at B=32 the kernels of one camera already fill the chip and
extra streams only add contention (tools/ab_step.py: 306 ms/step with 1 stream, 312 with 2, 321 with 4); the
weight-gradient side stream of ops_nn is what pays (327 -> 306 ms).
"""
import os

import torch

NUM = max(1, int(os.environ.get('XAS_CAM_STREAMS', '1')))   # measured on MI355X: 1 -> 306 ms/step, 2 -> 312, 4 -> 321
_cam = []
_book = [None]
_active = [False]          # True while camera streams are forked (BN bookkeeping must use the book stream)


def enabled():
    return NUM > 1 and torch.cuda.is_available()


def cam_stream(i):
    while len(_cam) < NUM:
        _cam.append(torch.cuda.Stream())
    return _cam[i % NUM]


def book_stream():
    if _book[0] is None:
        _book[0] = torch.cuda.Stream()
    return _book[0]


def forked():
    return _active[0]


class fork:
    """Context: camera streams may be used inside; on exit the main stream waits for all of them and for the
    bookkeeping stream."""

    def __enter__(self):
        self.main = torch.cuda.current_stream()
        _active[0] = enabled()
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


def to_main(*tensors):
    """Tell the caching allocator that tensors produced on a camera stream are consumed on the main stream."""
    if not enabled():
        return
    main = torch.cuda.current_stream()
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(main)
