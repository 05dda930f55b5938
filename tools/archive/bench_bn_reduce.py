"""Stand-alone rate of the batch-norm backward reduction (xas_bn_bwd_reduce) and of the apply pass on the layer shapes of
the detector at B = 32 x 8 images, per tuning value (column-reduce slab target, lean / full register build).
   python tools/bench_bn_reduce.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'x-as-supervision_amd')):
    sys.path.insert(0, p)
import torch                                             # noqa: E402
from xas_amd._lib import call, ptr, query               # noqa: E402

G = 8
SHAPES = [(256 * 64 * 64, 256, 'masked'), (256 * 64 * 64, 64, 'yfree'), (256 * 32 * 32, 512, 'masked'), (256 * 32 * 32, 128, 'yfree'),
          (256 * 16 * 16, 1024, 'masked'), (256 * 16 * 16, 256, 'yfree'), (256 * 8 * 8, 2048, 'masked')]


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


for tune in [int(t) for t in (sys.argv[1:] or ['0', '32768', '262144'])]:
    query('xas_set_tuning', tune)
    print('tune', tune)
    for M, C, form in SHAPES:
        x = torch.randn(M, C, device='cuda')
        dy = torch.randn(M, C, device='cuda')
        mean = torch.zeros(G, C, device='cuda'); var = torch.ones(G, C, device='cuda')
        gam = torch.ones(C, device='cuda'); bet = torch.zeros(C, device='cuda')
        sums = torch.empty(G, 2, C, device='cuda')
        ws = torch.empty(query('xas_bn_workspace_floats', M, C, G), device='cuda')
        mask = torch.randint(0, 16, (M * C // 4,), device='cuda', dtype=torch.uint8) if form == 'masked' else None
        dx = torch.empty_like(x)
        if form == 'masked':
            red = lambda: call('xas_bn_bwd_reduce', ptr(x), None, ptr(dy), ptr(mean), ptr(var), ptr(gam), ptr(bet), 1e-5, 1, M, C, G,
                               ptr(sums), ptr(ws), None, None, ptr(mask))
            app = lambda: call('xas_bn_bwd_apply', ptr(x), None, ptr(dy), ptr(mean), ptr(var), ptr(gam), ptr(bet), ptr(sums), 1e-5, 1, M, C, G,
                               float(M // G), ptr(dx), None, ptr(mask))
            rb, ab = 8.0 * M * C + M * C / 4, 12.0 * M * C + M * C / 4
        else:
            red = lambda: call('xas_bn_bwd_reduce', ptr(x), None, ptr(dy), ptr(mean), ptr(var), ptr(gam), ptr(bet), 1e-5, 1, M, C, G,
                               ptr(sums), ptr(ws), None, None, None)
            app = lambda: call('xas_bn_bwd_apply', ptr(x), None, ptr(dy), ptr(mean), ptr(var), ptr(gam), ptr(bet), ptr(sums), 1e-5, 1, M, C, G,
                               float(M // G), ptr(dx), None, None)
            rb, ab = 8.0 * M * C, 12.0 * M * C
        tr, ta = timed(red), timed(app)
        print('  M=%8d C=%4d %-6s reduce %7.1f us %5.2f TB/s | apply %7.1f us %5.2f TB/s' % (M, C, form, tr, rb / tr / 1e6, ta, ab / ta / 1e6))
        del x, dy, dx, mask
