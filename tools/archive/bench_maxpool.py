"""Stand-alone rate of the 3x3 stride-2 max pool (forward / backward) at the detector's size.   python tools/bench_maxpool.py [images]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'x-as-supervision_amd')):
    sys.path.insert(0, p)
import torch                                             # noqa: E402
from xas_amd import ops_nn                               # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(n, 64, 128, 128, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
y = ops_nn.maxpool3x3s2(x)
g = torch.randn_like(y)
ref = torch.nn.functional.max_pool2d(x.detach(), 3, 2, 1)
assert torch.equal(y.detach(), ref)
y.backward(g)
xr = x.detach().clone().requires_grad_(True)
torch.nn.functional.max_pool2d(xr, 3, 2, 1).backward(g)
print('max |dx - torch| = %.2e' % float((x.grad - xr.grad).abs().max()))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


with torch.no_grad():
    tf = timed(lambda: ops_nn.maxpool3x3s2(x))
fb = x.numel() * 4 + y.numel() * 5
print('forward  %7.1f us  %.2f TB/s' % (tf, fb / tf / 1e6))


def bwd():
    x.grad = None
    y2 = ops_nn.maxpool3x3s2(x)
    y2.backward(g)


tb = timed(bwd) - tf
bb = x.numel() * 4 + y.numel() * 5
print('backward %7.1f us  %.2f TB/s (forward + backward minus forward)' % (tb, bb / tb / 1e6))
