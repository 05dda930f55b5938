"""Geodesic weight maps on the GPU (reference: human_utils/common/utility/geodesic.py:14-54, scikit-fmm on the CPU,
two fast-marching solves per sample and camera).

`compute_geodesic_dis_batch` is the product path: one HIP launch produces the maps of a whole batch of device masks
(called by human_utils/dataloader/gpu_patch.py).  `compute_geodesic_dis` keeps the REFERENCE's positional signature
(`img [1,H,W]` numpy, `img_path`, parameter list, optional centres, `is_norm`) and return types, so that the reference's
own loader (`human_utils/dataloader/dataloader.py:13,80`) keeps working with the mirror in front of it on PYTHONPATH: it
runs the batch kernel on a batch of one.  `is_norm=False` (raw distances; no shipped config) and anything else this
module does not define fall through to the reference module.  So does every call from a DataLoader WORKER process
(train.py:278 starts `--worker 10` of them by fork after the parent has initialised the GPU: a forked child must never touch
the device - "Cannot re-initialize CUDA in forked subprocess") and every call in a process without a usable GPU: there the
reference's scikit-fmm implementation runs, exactly as without the mirror.  The GPU maps of a training run come from
`gpu_patch` on whole device batches, not from this per-sample entry.
"""
import ctypes
import os

import numpy as np
import torch

from xas_amd._lib import call, ptr, query


FMM_ORDER = int(os.environ.get('XAS_FMM_ORDER', '2'))   # 2: scikit-fmm's default scheme (skfmm.distance(m), geodesic.py:35,39); 1: the first-order maps of rounds 3-4


def compute_geodesic_dis_batch(mask, geodesic_param_list, centers=None, order=None):
    """mask [B,1,P,P] float device tensor (non-zero = foreground) -> (weights [B,1,P,P] float32, centres [B,2] or [B,n,2] int32 (x, y)).
    centers: optional [B,2] or [B,n,2] integer tensor of source pixels (the geodesic_pt_list joints); default: mask centroid.
    order: 1 / 2 = order of the upwind scheme of the two distance solves (default FMM_ORDER = 2, scikit-fmm's default)."""
    if mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[2] != mask.shape[3]:
        raise RuntimeError('compute_geodesic_dis_batch expects a [B,1,P,P] mask batch')
    order = FMM_ORDER if order is None else int(order)
    mask = mask.contiguous().float()
    B, _, P, _ = mask.shape
    out = torch.empty_like(mask)
    cen = torch.empty(B, 2, device=mask.device, dtype=torch.int32)
    ws = torch.empty(query('xas_geodesic_workspace_bytes', B, P), device=mask.device, dtype=torch.uint8)
    params = (ctypes.c_float * 5)(*[float(v) for v in geodesic_param_list])
    c = centers.to(device=mask.device, dtype=torch.int32).contiguous() if centers is not None else None
    nc = 1
    if c is not None and c.dim() == 3:                 # several sources per image
        if c.shape[0] != B or c.shape[2] != 2 or not 1 <= c.shape[1] <= 64:
            raise RuntimeError('compute_geodesic_dis_batch: centres must be [B, n, 2] with 1 <= n <= 64')
        nc = int(c.shape[1])
    call('xas_geodesic_weight_multi', ptr(mask), ptr(c), nc, order, ctypes.cast(params, ctypes.c_void_p), B, P, ptr(out), ptr(cen), ptr(ws))
    return out, (c if nc > 1 else cen)


def compute_centroid(mask):
    """[1,H,W] boolean mask -> int16 (x, y) centroid (geodesic.py:4-12)."""
    _, h, w = mask.shape
    grid = np.mgrid[0:h, 0:w]
    return np.array([np.sum(grid[1] * mask) / np.sum(mask), np.sum(grid[0] * mask) / np.sum(mask)]).astype(np.int16)


def _in_loader_worker():
    """True in a DataLoader worker process or in any forked child of a process that had initialised the GPU."""
    import torch.utils.data
    bad_fork = getattr(torch.cuda, '_is_in_bad_fork', None)          # (private helper of torch.cuda: present in 2.x)
    return torch.utils.data.get_worker_info() is not None or bool(bad_fork and bad_fork())


def compute_geodesic_dis(img, img_path, geodesic_param_list, centers=None, is_norm=True):
    """Reference signature (geodesic.py:14): img [1,H,W] numpy mask -> (weight map [1,H,W], centres [n,2] int16)."""
    if (not is_norm or img.shape[-1] != img.shape[-2] or (centers is not None and not 1 <= len(centers) <= 64) or _in_loader_worker()
            or not torch.cuda.is_available()):
        return __getattr__('compute_geodesic_dis')(img, img_path, geodesic_param_list, centers, is_norm)
    dev = torch.device('cuda', torch.cuda.current_device())
    m = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).to(dev)[None]
    # (a centre may carry a depth column - the reference indexes center[0], center[1] only, geodesic.py:23-24,30-31)
    c = None if centers is None else torch.as_tensor(np.ascontiguousarray(np.asarray(centers)[:, :2], dtype=np.int32))
    if c is not None and len(c) > 1:
        c = c[None]                                    # [1, n, 2]: several sources
    out, cen = compute_geodesic_dis_batch(m, geodesic_param_list, c)
    return out[0].cpu().numpy(), cen.cpu().numpy().astype(np.int16).reshape(-1, 2)


from xas_amd._next import fallthrough as _fallthrough  # noqa: E402

__getattr__ = _fallthrough(__name__, __file__)
