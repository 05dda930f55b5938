#!/usr/bin/env python3
"""GPU input pipeline throughput: one benchmark batch = 32 samples x 4 cameras of 1000 x 1002 frames
(Human3.6M size) -> 256 x 256 patches, masks, geodesic weight maps.  Frames are already resident as uint8 in HBM for the
kernel-only figure; the host-inclusive figure adds packing + upload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'x-as-supervision_amd')]
import numpy as np
import torch
from human_utils.dataloader.gpu_patch import generate_patch_batch
rng = np.random.Generator(np.random.PCG64(0))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H, W = 1002, 1000
frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(8)] * (B // 8)
yy, xx = np.mgrid[0:H, 0:W]
m = (((yy - 500) / 330.0) ** 2 + ((xx - 500) / 130.0) ** 2 < 1).astype(np.uint8) * 255
masks = [m] * B
samples = [{'center_x': 500.0, 'center_y': 500.0, 'width': 900.0, 'height': 900.0, 'rot': 0.0, 'joints_3d': rng.uniform(0, 1000, (18, 3)),
            'joints_3d_vis': np.ones((18, 3)), 'flip_pairs': []} for _ in range(B)]
dev = torch.device('cuda')
for _ in range(2):
    out = generate_patch_batch(samples, frames, masks, 256, 256, 2000, [0, 0, 0], [255, 255, 255], dev)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3):
    out = generate_patch_batch(samples, frames, masks, 256, 256, 2000, [0, 0, 0], [255, 255, 255], dev)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 3
print('host-inclusive (pack + upload %.0f MB + kernels): %.1f ms per %d images = %.0f images/s' % (B * H * W * 4 / 1e6, dt * 1e3, B, B / dt))
