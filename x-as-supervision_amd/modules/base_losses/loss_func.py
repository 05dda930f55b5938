"""Losses of the training step (reference: modules/base_losses/loss_func.py:4-76).

The two mask losses (2 M pixels each, 8 per step) are fused HIP reductions; the LSGAN terms and the
min-over-hypotheses symmetry / supervision losses used by the model are single fused kernels too
(`compute_symmetry_min`, `compute_supervision_min`); the per-function forms below keep the reference API.
"""
import torch

from xas_amd import ops_misc

_LIMB_FAR = [16, 15, 13, 12, 3, 2, 6, 5]
_LIMB_NEAR = [15, 14, 12, 11, 2, 1, 5, 4]


def compute_mask_reconstruction_loss(mask, gt, weight=None, use_clip=False):
    """MSE(mask, gt) with optional clip (mask > 0.1) and geodesic weight map (loss_func.py:4-16).
    Returns the SCALAR the trainer reduces to: with weight=None and use_clip=True the reference returns
    the tensor mse * clip and takes .mean() at train.py:182; the value is identical."""
    return ops_misc.mask_loss(mask, gt, weight, use_clip)


def compute_bone_sym_loss(keypoints):
    limb = (keypoints[:, _LIMB_FAR] - keypoints[:, _LIMB_NEAR]).norm(dim=2) * 1e-3
    return ((limb[:, 0::2] - limb[:, 1::2]) ** 2).mean()


def compute_kp_sym_loss(keypoints, is_3D=True):
    mid = (keypoints[:, [11, 1]] + keypoints[:, [14, 4]]) / 2
    anchor = keypoints[:, [-1, 0]]
    if is_3D:
        return ((mid * 1e-3 - anchor * 1e-3) ** 2).mean()
    return ((mid - anchor) ** 2).mean()


def compute_supervision(keypoint, keypoint_gt, feature_shape=None, mode='mean'):
    if feature_shape is not None:
        scale = keypoint.new_tensor([feature_shape[0] - 1, feature_shape[1] - 1] +
                                    ([feature_shape[2] - 1] if keypoint.shape[-1] == 3 else []))
        xy = (keypoint[..., :2] + 1) / 2.0
        keypoint = torch.cat([xy, keypoint[..., 2:]], dim=-1) * scale
    err = (keypoint - keypoint_gt) ** 2
    if mode == 'mean':
        return err.mean()
    if mode == 'sum':
        return err.sum() / keypoint.shape[0]
    return err


def _lsgan(logits, target):
    if logits.dim() not in (2, 3):
        raise ValueError('Invalid dimension of logits')
    return ops_misc.lsgan_term(logits, target)    # fused: per-sample min over the hypothesis axis, batch mean


def compute_disc_loss(pred_logits, gt_logits):
    if gt_logits is None:
        return _lsgan(pred_logits, 1.0)
    return 0.5 * _lsgan(gt_logits, 1.0) + 0.5 * _lsgan(pred_logits, 0.0)


def compute_supervision_min(pred, gt):
    """min over hypotheses h of compute_supervision(pred[:, h], gt): one fused kernel (model.py:158-162)."""
    return ops_misc.supervision_min(pred, gt)


def compute_symmetry_min(world, w_bone, w_kp, kps=None, w_kp2d=None):
    """min over hypotheses of w_bone*bone_sym + w_kp*kp_sym [+ 100*w_kp2d*kp_sym_2d]: one fused kernel
    (model.py:104-114)."""
    return ops_misc.symmetry_min(world, w_bone, w_kp, kps if w_kp2d is not None else None, w_kp2d or 0.0)


# names this mirror does not replace resolve, lazily, to the reference module behind it on sys.path
from xas_amd._next import fallthrough as _fallthrough  # noqa: E402

__getattr__ = _fallthrough(__name__, __file__)
