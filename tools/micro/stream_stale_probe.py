#!/usr/bin/env python3
"""Platform probe, no kernel of this repository: does a producer -> consumer chain of small stock-PyTorch kernels on ONE
stream ever read stale data while other streams keep the chip busy with unrelated stock kernels?

main stream : per iteration, a chain over freshly allocated small tensors - scatter-style writes (index_copy of 12-byte
              rows, as the soft-argmax head writes joints), a gather-style read, elementwise ops - reduced to a checksum that
              must equal the one computed with the other streams idle.
other streams: large elementwise / matmul kernels on their own tensors, launched continuously.
usage: python tools/micro/stream_stale_probe.py [iterations] [n_other_streams]
"""
import sys

import torch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n_other = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(1)
B, H, K = 128, 3, 18
src = torch.randn(B, H, K, 3, device=dev, generator=g)
perm = torch.randperm(B * H * K, device=dev, generator=g)
cam = torch.randn(B, 3, 3, device=dev, generator=g)


def chain():
    # producer: rows written in a scattered order (different workgroups write neighbouring 12-byte rows of one cache line)
    kps = torch.empty(B * H * K, 3, device=dev)
    kps.index_copy_(0, perm, src.reshape(-1, 3)[perm])
    kps = kps.view(B, H, K, 3)
    # consumer 1: a per-sample 3x3 transform (reads the rows just written), consumer 2: gather + reduction
    w = torch.einsum('bhkc,bdc->bhkd', kps, cam)
    z = (w - w[:, 0:1]) * 1e-3
    loss = (z[:, :, [16, 15, 13, 12]] - z[:, :, [15, 14, 12, 11]]).pow(2).sum((-1, -2)).min(1).values.mean()
    return torch.stack([loss, w.double().sum().float(), kps.double().sum().float()])


others = [torch.cuda.Stream() for _ in range(n_other)]
big = [torch.randn(4096, 4096, device=dev, generator=g) for _ in range(n_other)]
act = [torch.randn(64, 256, 64, 64, device=dev, generator=g) for _ in range(n_other)]
ref = chain()
torch.cuda.synchronize()
assert torch.equal(chain(), ref)
torch.cuda.synchronize()
bad = 0
for it in range(iters):
    for s, m, a in zip(others, big, act):
        with torch.cuda.stream(s):
            t = torch.relu(a * 1.0001 + 0.5)              # fresh 268 MB result, freed at once: the pool's blocks are recycled
            u = m @ m
            del t, u
    got = chain()
    if it % 50 == 49:
        torch.cuda.synchronize()
    if not torch.equal(got, ref):
        bad += 1
        print('iteration %d: checksum differs: %s vs %s' % (it, got.tolist(), ref.tolist()), flush=True)
torch.cuda.synchronize()
print('stock-PyTorch probe: %d of %d chains differed with %d other streams busy' % (bad, iters, n_other))
