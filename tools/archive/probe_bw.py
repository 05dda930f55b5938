#!/usr/bin/env python3
"""Device-to-device copy bandwidth every 0.5 s for ~12 s (is the GPU quiet after the previous process?)."""
import time, torch
a = torch.empty(2 ** 28, dtype=torch.float32, device='cuda')
b = torch.empty_like(a)
t0 = time.time()
while time.time() - t0 < 12:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    print('t=%.1fs  %.2f TB/s' % (time.time() - t0, 4 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12), flush=True)
    time.sleep(0.5)
