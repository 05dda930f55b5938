"""Pose metrics on the MI355X kernel (reference: metrics.py).  Same names, arguments and return types as the
reference (numpy arrays on the host); `pose_errors` is the device-resident form the evaluation loop uses so that a
batch costs one launch and no host round trip per alignment."""
import numpy as np
import torch

from xas_amd import ops_eval

_ALIGN = {'none': 0, 'scale': 1, 'procrustes': 2}


def _dev(a):
    if isinstance(a, torch.Tensor):
        t = a
    else:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError('metrics run on the MI355X kernel (xas_pose_metrics); no GPU is visible and there is no CPU fallback')
        t = t.cuda()
    return t.float()


def _align_id(alignment):
    if alignment not in _ALIGN:
        raise ValueError(f'Invalid value for alignment: {alignment}')
    return _ALIGN[alignment]


def pose_errors(pred, gt, mask=None, in_div=1.0, pck_align='none', threshold=0.15, want=('err', 'pck', 'auc_hits')):
    """Device tensors: err [3,N,K] (none, scale, procrustes), pck [N,K], auc_hits [N,31]."""
    return ops_eval.pose_metrics(_dev(pred), _dev(gt), mask, in_div=in_div, pck_align=_align_id(pck_align),
                                 pck_threshold=threshold, want=want)


def compute_similarity_transform(source_points, target_points):
    """Similarity (scale, proper rotation, translation) that best maps source [N,3] onto target [N,3]; returns the
    transformed source points (metrics.py:5-62)."""
    assert target_points.shape[0] == source_points.shape[0]
    assert target_points.shape[1] == 3 and source_points.shape[1] == 3
    out = ops_eval.pose_metrics(_dev(source_points)[None], _dev(target_points)[None], want=('aligned',))
    return out['aligned'][1, 0].cpu().numpy()


def keypoint_mpjpe(pred, gt, mask, alignment='none'):
    """Per-joint position error [N,K] (x mask) without / after scale / after procrustes alignment
    (metrics.py:65-118)."""
    mask = np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask)
    assert mask.any()
    a = _align_id(alignment)
    return pose_errors(pred, gt, mask, want=('err',))['err'][a].cpu().numpy()


def keypoint_3d_pck(pred, gt, mask, alignment='none', threshold=0.15):
    """[N,K] of 0 / 100: joint error under `threshold` (metrics.py:121-175)."""
    mask = np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask)
    assert mask.any()
    return pose_errors(pred, gt, mask, pck_align=alignment, threshold=threshold, want=('pck',))['pck'].cpu().numpy()


def keypoint_3d_auc(pred, gt, mask, alignment='none'):
    """Area under the PCK curve for 31 thresholds in [0, 0.15] (metrics.py:178-244)."""
    mask = np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask)
    assert mask.any()
    hits = pose_errors(pred, gt, mask, pck_align=alignment, want=('auc_hits',))['auc_hits']
    n, k = mask.shape
    return float((hits.sum(dim=0).double() / (n * k)).mean().item() * 100)


def keypoint_pckh(pred, gt, head_size, PCKh_thred=0.5):
    """PCKh per sample (metrics.py:247-253)."""
    error = torch.linalg.norm(pred - gt, ord=2, dim=-1) / head_size.unsqueeze(-1)
    return (error < PCKh_thred).float().mean(dim=-1) * 100


# names this mirror does not replace resolve, lazily, to the reference module behind it on sys.path
from xas_amd._next import fallthrough as _fallthrough  # noqa: E402

__getattr__ = _fallthrough(__name__, __file__)
