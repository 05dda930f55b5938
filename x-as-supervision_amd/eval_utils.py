"""Evaluation helpers on the MI355X kernels (reference: eval_utils.py).  The plotting helpers of the reference
file (show3Dpose, plot_*) are visualisation and not rebuilt."""
import torch

from xas_amd import ops_eval

SWITCH_LIST = [(1, 4), (2, 5), (3, 6), (14, 11), (15, 12), (16, 13)]


def switch_points(points, gt, switch_all=False, switch_list=SWITCH_LIST):
    """Resolve the left/right ambiguity against the ground truth: [B,K,C] x2 -> (points', is_trans [B,K,1]) with
    the mirrored joint kept where its 2-D L1 error is strictly smaller (eval_utils.py:7-30).  One launch."""
    out = ops_eval.eval_select(points.unsqueeze(1), gt, pairs=switch_list, mode='confident', switch_all=switch_all,
                               gt_normalised=True, want=('sel3d', 'swapped'))
    is_trans = out['swapped']
    if switch_all:
        is_trans = is_trans[:, :1]                     # one decision per sample: [B,1,1] as the reference returns
    return out['sel3d'], is_trans


def per_act_mse(pred, gt):
    """Mean 2-D joint distance in the unit square, [B,K,2] x2 -> [B] (eval_utils.py:32-43)."""
    return ops_eval.eval_select(pred.unsqueeze(1), gt, pairs=(), mode='confident', gt_normalised=True,
                                want=('err2d',))['err2d']


def cal_per_class_error_(record_table, count_table):
    """Per-action means, their average and the 6-action 'select' average (eval_utils.py:45-59)."""
    full_err, select_err = 0.0, 0.0
    for k in record_table.keys():
        record_table[k] /= (count_table[k] + 1e-8)
        full_err += record_table[k]
        if k in ['Waiting', 'Posing', 'Greeting', 'Directions', 'Discussion', 'Walking']:
            select_err += record_table[k]
    return full_err / len(record_table), select_err / 6


def cal_per_class_error(record_table, count_table, multi=False):
    if not multi:
        return cal_per_class_error_(record_table, count_table)
    full_err, select_err = {}, {}
    for metric in record_table.keys():
        full_err[metric], select_err[metric] = cal_per_class_error_(record_table[metric], count_table[metric])
    return full_err, select_err


# names this mirror does not replace resolve, lazily, to the reference module behind it on sys.path
from xas_amd._next import fallthrough as _fallthrough  # noqa: E402

__getattr__ = _fallthrough(__name__, __file__)
