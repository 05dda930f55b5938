"""Geodesic weight maps on the GPU (reference: human_utils/common/utility/geodesic.py:14-54, scikit-fmm on the CPU,
two fast-marching solves per sample and camera).  One HIP launch produces the maps of a whole batch."""
import ctypes

import torch

from xas_amd._lib import call, ptr, query


def compute_geodesic_dis(mask, geodesic_param_list, centers=None):
    """mask [B,1,P,P] float device tensor (non-zero = foreground) -> (weights [B,1,P,P] float32, centres [B,2] int32 (x, y)).
    centers: optional [B,2] integer tensor of source pixels (geodesic_pt_list joints); default: mask centroid."""
    if mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[2] != mask.shape[3]:
        raise RuntimeError('compute_geodesic_dis expects a [B,1,P,P] mask batch')
    mask = mask.contiguous().float()
    B, _, P, _ = mask.shape
    out = torch.empty_like(mask)
    cen = torch.empty(B, 2, device=mask.device, dtype=torch.int32)
    ws = torch.empty(query('xas_geodesic_workspace_bytes', B, P), device=mask.device, dtype=torch.uint8)
    params = (ctypes.c_float * 5)(*[float(v) for v in geodesic_param_list])
    c = centers.to(device=mask.device, dtype=torch.int32).contiguous() if centers is not None else None
    call('xas_geodesic_weight', ptr(mask), ptr(c), ctypes.cast(params, ctypes.c_void_p), B, P, ptr(out), ptr(cen), ptr(ws))
    return out, cen
