"""Evaluation-path ops over the C ABI: hypothesis selection, triangulation, pose metrics (no autograd: eval runs
under torch.no_grad, eval.py:393).  Everything returns device tensors; nothing here synchronises with the host."""
import torch

from ._lib import call, ptr
from .ops_head import GEO_NORM, GEO_PATCH

GEO_IMAGE = 8
EVAL_CONFIDENT, EVAL_SWITCH_ALL, EVAL_GT_NORMALISED = 1, 2, 4
SWITCH_PAIRS = ((1, 4), (2, 5), (3, 6), (14, 11), (15, 12), (16, 13))     # eval_utils.py:8

_perm_cache = {}


def switch_perm(num_kp, pairs, device):
    key = (num_kp, tuple(tuple(p) for p in pairs), str(device))
    if key not in _perm_cache:
        perm = list(range(num_kp))
        for a, b in pairs:
            perm[a], perm[b] = b, a
        _perm_cache[key] = torch.tensor(perm, dtype=torch.int32, device=device)
    return _perm_cache[key]


def _f32(t):
    return t.detach().contiguous().float()


def eval_select(kps, joints, pairs=SWITCH_PAIRS, image_size=256.0, mode='best', switch_all=False, gt_normalised=False,
                want=('sel3d', 'sel2d', 'err2d', 'swapped')):
    """kps [B,Hy,K,C], joints [B,K,C] -> dict of the requested outputs (eval.py:117-148 for one camera)."""
    if mode not in ('best', 'confident'):
        raise ValueError('Unknown mode: {}'.format(mode))
    kps, joints = _f32(kps), _f32(joints)
    B, Hy, K, C = kps.shape
    if joints.shape != (B, K, C):
        raise RuntimeError('eval_select: joints %s do not match kps %s' % (tuple(joints.shape), tuple(kps.shape)))
    dev = kps.device
    flags = (EVAL_CONFIDENT if (mode == 'confident' or Hy == 1) else 0) | (EVAL_SWITCH_ALL if switch_all else 0) | \
        (EVAL_GT_NORMALISED if gt_normalised else 0)
    out = {}
    if 'sel3d' in want:
        out['sel3d'] = torch.empty(B, K, C, device=dev)
    if 'sel2d' in want:
        out['sel2d'] = torch.empty(B, K, 2, device=dev)
    if 'err2d' in want:
        out['err2d'] = torch.empty(B, device=dev)
    if 'swapped' in want:
        out['swapped'] = torch.empty(B, K, 1, device=dev, dtype=torch.uint8)
    call('xas_eval_select', ptr(kps), ptr(joints), ptr(switch_perm(K, pairs, dev)), B, Hy, K, C, float(image_size), flags,
         ptr(out.get('sel3d')), ptr(out.get('sel2d')), ptr(out.get('err2d')), ptr(out.get('swapped')))
    if 'swapped' in out:
        out['swapped'] = out['swapped'].bool()
    return out


def patch_to_image(kps, trans_image, pelvis, image_size=256.0, rect_width=2000.0, is_norm=True):
    """[B,K,3] normalised patch coordinates -> image (u px, v px, depth mm).  modules/util.py:61-83."""
    kps = _f32(kps).unsqueeze(1)
    B, _, K, _ = kps.shape
    ti, pv = _f32(trans_image), _f32(pelvis)
    out = torch.empty_like(kps)
    call('xas_patch_to_world_fwd', ptr(kps), ptr(ti), None, ptr(pv), None, None, B, 1, K, float(image_size),
         float(rect_width), (GEO_NORM if is_norm else 0) | GEO_PATCH | GEO_IMAGE, ptr(out))
    return out.squeeze(1)


def projection_matrix(k_mat, rot_world, trans_world):
    """P [B,3,4] = K [R | t].  modules/util.py:188."""
    km, rw, tw = _f32(k_mat), _f32(rot_world), _f32(trans_world)
    P = torch.empty(km.shape[0], 3, 4, device=km.device)
    call('xas_projection_matrix', ptr(km), ptr(rw), ptr(tw), km.shape[0], ptr(P))
    return P


def triangulate_dlt(points, pmat):
    """points [B,V,K,3] (u, v, weight), pmat [B,V,3,4] -> [B,K,4] (X/w, mean weight).  modules/util.py:198-230."""
    points, pmat = _f32(points), _f32(pmat)
    B, V, K, _ = points.shape
    if pmat.shape != (B, V, 3, 4):
        raise RuntimeError('triangulate_dlt: projection matrices %s do not match points %s' % (tuple(pmat.shape), tuple(points.shape)))
    out = torch.empty(B, K, 4, device=points.device)
    call('xas_triangulate_dlt', ptr(points), ptr(pmat), B, V, K, ptr(out))
    return out


def pose_metrics(pred, gt, mask=None, in_div=1.0, pck_align=0, pck_threshold=0.15,
                 want=('err', 'pck', 'auc_hits')):
    """pred, gt [N,K,3] -> dict: err [3,N,K] (none / scale / procrustes), aligned [2,N,K,3], pck [N,K],
    auc_hits [N,31] int32 (metrics.py:5-244)."""
    pred, gt = _f32(pred), _f32(gt)
    N, K, _ = pred.shape
    if gt.shape != pred.shape:
        raise RuntimeError('pose_metrics: pred %s and gt %s differ' % (tuple(pred.shape), tuple(gt.shape)))
    dev = pred.device
    m = None
    if mask is not None:
        m = torch.as_tensor(mask, device=dev).to(torch.uint8).contiguous()
    out = {}
    if 'err' in want:
        out['err'] = torch.empty(3, N, K, device=dev)
    if 'aligned' in want:
        out['aligned'] = torch.empty(2, N, K, 3, device=dev)
    if 'pck' in want:
        out['pck'] = torch.empty(N, K, device=dev)
    if 'auc_hits' in want:
        out['auc_hits'] = torch.empty(N, 31, device=dev, dtype=torch.int32)
    call('xas_pose_metrics', ptr(pred), ptr(gt), ptr(m), N, K, float(in_div), int(pck_align), float(pck_threshold),
         ptr(out.get('err')), ptr(out.get('aligned')), ptr(out.get('pck')), ptr(out.get('auc_hits')))
    return out
