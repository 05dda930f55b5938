"""Oracle: loss functions.  TEST INFRASTRUCTURE ONLY.
Follows modules/base_losses/loss_func.py:4-76."""
import torch

_BONE_A = [16, 15, 13, 12, 3, 2, 6, 5]   # loss_func.py:20
_BONE_B = [15, 14, 12, 11, 2, 1, 5, 4]


def mask_recon(mask, gt, weight=None, use_clip=False):
    """loss_func.py:4-16.  NB: weight=None & use_clip=True returns a TENSOR
    (scalar MSE x clip mask), reduced later by .mean() at train.py:182."""
    sq = (mask - gt) ** 2
    loss = sq.mean() if weight is None else sq
    if use_clip:
        loss = loss * (mask > 0.1).to(mask.dtype)
    if weight is not None:
        loss = (loss * weight).mean()
    return loss


def bone_sym(kp):
    """loss_func.py:18-25: left/right limb lengths (mm * 1e-3) should agree."""
    bone = (kp[:, _BONE_A] - kp[:, _BONE_B]).norm(dim=2) * 1e-3
    return ((bone[:, 0::2] - bone[:, 1::2]) ** 2).mean()


def kp_sym(kp, is_3d=True):
    """loss_func.py:27-35: shoulder/hip mid-points vs thorax/root."""
    center = (kp[:, [11, 1]] + kp[:, [14, 4]]) / 2
    target = kp[:, [-1, 0]]
    s = 1e-3 if is_3d else 1.0
    return ((center * s - target * s) ** 2).mean()


def supervision(kp, kp_gt):
    """loss_func.py:38-52 with feature_shape=None, mode='mean'."""
    return ((kp - kp_gt) ** 2).mean()


def _lsgan_term(logits, target):
    e = (logits - target) ** 2
    if logits.dim() == 2:
        return e.mean()
    if logits.dim() == 3:
        return e.min(dim=1)[0].mean()
    raise ValueError('Invalid dimension of logits')


def disc_loss(pred_logits, gt_logits):
    """loss_func.py:54-76 (LSGAN, per-sample min over the hypothesis axis)."""
    if gt_logits is None:
        return _lsgan_term(pred_logits, 1.0)
    return 0.5 * _lsgan_term(gt_logits, 1.0) + 0.5 * _lsgan_term(pred_logits, 0.0)
