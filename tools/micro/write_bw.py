#!/usr/bin/env python3
"""What HBM delivers for WRITE-heavy streams on this part (stock PyTorch kernels, large tensors): pure fill, copy (1 read :
1 write), and a 1 read : 4 writes pattern (the traffic mix of the P -> 4P 1x1 convolutions).  usage: python tools/micro/write_bw.py"""
import torch


def timed(fn, nbytes, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return ms, nbytes / ms / 1e9


def main():
    n = 1 << 29                                   # 2 GiB of fp32
    x = torch.randn(n // 4, device='cuda')
    y = torch.empty(n, device='cuda')
    z = torch.empty(n, device='cuda')
    print('fill 2 GiB            : %.3f ms  %.2f TB/s' % timed(lambda: y.fill_(1.0), n * 4))
    print('copy 2 GiB -> 2 GiB   : %.3f ms  %.2f TB/s' % timed(lambda: z.copy_(y), 2 * n * 4))
    xv, yv = x.view(-1, 1, 64), y.view(-1, 4, 64)
    print('1 read : 4 writes     : %.3f ms  %.2f TB/s' % timed(lambda: yv.copy_(xv.expand(-1, 4, -1)), (n + n // 4) * 4))
    print('4 reads : 1 write (sum): %.3f ms  %.2f TB/s' % timed(lambda: torch.sum(yv, dim=1, out=x.view(-1, 64)), (n + n // 4) * 4))


if __name__ == '__main__':
    main()
