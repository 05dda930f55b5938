"""Soft-argmax head, patch->world geometry and line-mask renderer as autograd ops over the C ABI."""
import torch

from . import _lib
from ._lib import call, ptr, query

HEAD_STATS = 16
peak_probe = None        # measurement hook (tests): called with the int64 depth-peak indices [B,K,Hy] of every head forward
GEO_NORM, GEO_MONO, GEO_PATCH = 1, 2, 4


def _nhwc_storage(logits):
    """[B,C,H,W] logical tensor -> the same tensor with NHWC (channels_last) storage."""
    if logits.dim() != 4:
        raise RuntimeError('logits must be [B, K*D, H, W]')
    if not logits.is_contiguous(memory_format=torch.channels_last):
        logits = logits.contiguous(memory_format=torch.channels_last)
    return logits


class _SoftArgmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, num_kp, num_hypo, neighbor, groups):
        from . import ops_nn
        logits_in = logits
        logits_tail = _nhwc_storage(logits)
        logits = ops_nn._full(logits_tail)               # prefix pass (ops_nn): prefix + graph images in one launch
        B, C, H, W = logits.shape
        D = C // num_kp
        if not (C == num_kp * D and D == H == W):
            raise RuntimeError('soft-argmax head needs D == H == W (got C=%d K=%d H=%d W=%d)' % (C, num_kp, H, W))
        dev = logits.device
        kps = torch.empty(B, num_hypo, num_kp, 3, device=dev, dtype=torch.float32)
        z_idx = torch.empty(B, num_kp, num_hypo, device=dev, dtype=torch.int64)
        dmap = torch.empty(groups, num_kp, D, device=dev, dtype=torch.float32)
        stats = torch.empty(B, num_kp, HEAD_STATS, device=dev, dtype=torch.float32)
        pre = getattr(logits_in, '_xas_head', None)      # first-pass records from the final convolution's epilogue (ops_nn._Conv2d)
        if pre is not None and pre[2] == logits_in._version and pre[0].shape[0] == B and pre[0].shape[2] == num_kp:
            ops_nn.head_stats['fused'] += 1
            call('xas_head_softargmax_from_partials', ptr(pre[0]), B, num_kp, D, pre[1], num_hypo, neighbor, ptr(kps), ptr(z_idx),
                 ptr(dmap), groups, ptr(stats))
        else:
            ops_nn.head_stats['separate'] += 1
            ws = torch.empty(query('xas_head_workspace_floats', B, num_kp, D), device=dev, dtype=torch.float32)
            call('xas_head_softargmax_fwd', ptr(logits), B, num_kp, D, num_hypo, neighbor, ptr(kps), ptr(z_idx),
                 ptr(dmap), groups, ptr(stats), ptr(ws))
        if peak_probe is not None:
            peak_probe(z_idx.clone())
        s = B - logits_tail.shape[0]                     # images of the no-grad prefix (0 outside a prefix pass)
        kps_prefix = kps[:s] if s else kps[:0]
        kps, z_idx_all = (kps[s:], z_idx) if s else (kps, z_idx)
        ctx.save_for_backward(logits_tail, stats[s:] if s else stats, z_idx[s:] if s else z_idx)
        ctx.cfg = (num_kp, D, num_hypo, neighbor)
        ctx.mark_non_differentiable(z_idx_all, dmap, kps_prefix)
        return kps, dmap, z_idx_all, kps_prefix

    @staticmethod
    def backward(ctx, g_kps, _g_dmap, _g_idx, _g_prefix=None):
        logits, stats, z_idx = ctx.saved_tensors
        K, D, Hy, nb = ctx.cfg
        B = logits.shape[0]
        g_kps = g_kps.contiguous()
        grad = torch.empty_like(logits)          # keeps the channels_last strides
        coef = torch.empty(B * K * (4 + D), device=logits.device, dtype=torch.float32)
        from . import ops_nn
        slot = ops_nn.grad_amax_slot(logits.device)          # max |grad|: the final conv's f16x3 gradient launches scale dy with it
        call('xas_head_softargmax_bwd_amax', ptr(logits), ptr(stats), ptr(z_idx), ptr(g_kps), B, K, D, Hy, nb,
             ptr(grad), ptr(coef), ptr(slot))
        if slot is not None:
            ops_nn.tag_grad_amax(grad, slot)
        return grad, None, None, None, None


def softargmax_multi(logits, num_kp, num_hypo, neighbor_size, groups=1):
    """-> kps [B,num_hypo,K,3], depth_prob_map [K,D] ([groups,K,D] for groups > 1: first sample of each sub-batch),
    z_idx [B,K,num_hypo] int64  (keypoint_detector_integral_multi.py:66-88)."""
    kps, dmap, idx, prefix = _SoftArgmax.apply(logits, num_kp, num_hypo, neighbor_size, groups)
    _last_prefix[0] = prefix
    return kps, (dmap[0] if groups == 1 else dmap), idx


_last_prefix = [None]          # joints of the no-grad prefix images of the latest call (prefix pass: KPDetector3DMulti.forward_groups)


def softargmax_single(logits, num_kp, groups=1):
    """-> kps [B,1,K,3], depth_prob_map [K,D] ([groups,K,D] for groups > 1)  (keypoint_detector_integral.py:45-65)."""
    kps, dmap, _, prefix = _SoftArgmax.apply(logits, num_kp, 1, 0, groups)
    _last_prefix[0] = prefix
    return kps, (dmap[0] if groups == 1 else dmap)


class _PatchToWorld(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kps, ti, km, pv, rw, tw, image_size, rect_width, flags):
        kps = kps.contiguous()
        B, Hy, K, _ = kps.shape
        cams = [t.contiguous().float() for t in (ti, km, pv, rw, tw)]
        world = torch.empty_like(kps)
        call('xas_patch_to_world_fwd', ptr(kps), *[ptr(c) for c in cams], B, Hy, K, float(image_size),
             float(rect_width), flags, ptr(world))
        ctx.save_for_backward(kps, *cams)
        ctx.cfg = (float(image_size), float(rect_width), flags)
        return world

    @staticmethod
    def backward(ctx, gw):
        kps, *cams = ctx.saved_tensors
        B, Hy, K, _ = kps.shape
        S, rect, flags = ctx.cfg
        gk = torch.empty_like(kps)
        call('xas_patch_to_world_bwd', ptr(kps), ptr(gw.contiguous()), *[ptr(c) for c in cams], B, Hy, K, S, rect,
             flags, ptr(gk))
        return (gk,) + (None,) * 8


def patch_to_world(kps, trans_image, k_mat, pelvis, rot_world, trans_world, image_size=256, rect_width=2000.0,
                   is_norm=True, mono=False, patch=True):
    """kps [B,K,3] or [B,Hy,K,3] -> world coordinates of the same shape (modules/util.py:128-152)."""
    squeeze = kps.dim() == 3
    if squeeze:
        kps = kps.unsqueeze(1)
    flags = (GEO_NORM if is_norm else 0) | (GEO_MONO if mono else 0) | (GEO_PATCH if patch else 0)
    out = _PatchToWorld.apply(kps, trans_image, k_mat, pelvis, rot_world, trans_world, image_size, rect_width, flags)
    return out.squeeze(1) if squeeze else out


_link_cache = {}


def _links(parents, children, device):
    key = (tuple(parents), tuple(children), str(device))
    if key not in _link_cache:
        _link_cache[key] = (torch.tensor(list(parents), dtype=torch.int32, device=device),
                            torch.tensor(list(children), dtype=torch.int32, device=device))
    return _link_cache[key]


class _DrawLinesMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kps2d, S, parents, children, body_width):
        kps2d = kps2d.contiguous()
        B, K, _ = kps2d.shape
        L = len(parents)
        par, chi = _links(parents, children, kps2d.device)
        fine = 0
        if L >= 21:                       # modules/util.py:50-53
            for l in (11, 12, 14, 15):
                fine |= 1 << l
        mask = torch.empty(B, 1, S, S, device=kps2d.device, dtype=torch.float32)
        call('xas_draw_lines_max_fwd', ptr(kps2d), K * 2, 2, B, K, ptr(par), ptr(chi), L, fine, float(body_width), S,
             ptr(mask))
        ctx.save_for_backward(kps2d, par, chi)
        ctx.cfg = (S, L, fine, float(body_width))
        return mask

    @staticmethod
    def backward(ctx, gmask):
        kps2d, par, chi = ctx.saved_tensors
        S, L, fine, width = ctx.cfg
        B, K, _ = kps2d.shape
        nblk = query('xas_lines_nblk', S)
        partial = torch.empty(B * nblk * K * 2, device=kps2d.device, dtype=torch.float32)
        g = torch.empty_like(kps2d)
        call('xas_draw_lines_max_bwd', ptr(kps2d), K * 2, 2, B, K, ptr(par), ptr(chi), L, fine, width, S,
             ptr(gmask.contiguous()), ptr(partial), ptr(g))
        return g, None, None, None, None


def draw_lines_max(kps2d, image_size, parents, children, body_width):
    """[B,K,2] -> max over line heat-maps [B,1,S,S]  (modules/util.py:21-59 + modules/model.py:94)."""
    return _DrawLinesMax.apply(kps2d, int(image_size), parents, children, body_width)
