#!/usr/bin/env python3
"""torchrun entry of the MI355X-native evaluation (reference: eval.py:1-408, same CLI and result file).

    torchrun --nproc-per-node=N eval.py --config config/HM36_Multi_SurS1.yaml --checkpoint ckpt.pth.tar
             [--batch_size B --worker W --multi_hypo best|confident] [--synthetic STEPS]

One evaluation batch = the detector in eval mode on every camera, then ON THE DEVICE: left/right switch +
hypothesis selection + 2-D error (one launch per camera), patch->world of every view, DLT triangulation of all
joints (one launch), MPJPE / N-MPJPE / P-MPJPE / PCK / AUC of every prediction set (one launch each), and ONE
host transfer of the per-sample numbers that the per-action tables (host dictionaries, as in the reference)
accumulate.  The reference does the same work with per-hypothesis tensor chains, a batched fp32 SVD and numpy
SVDs on the host (eval.py:111-204, metrics.py).  TensorBoard pose images (eval.py:150-156,172-202) are
visualisation and are not rebuilt.
"""
import copy
import os
from argparse import ArgumentParser

import numpy as np

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (RCCL between processes) on this driver
import torch
import yaml
from torch.distributed import destroy_process_group, init_process_group

from metrics import pose_errors
from modules.keypoint_detector_integral import KPDetector3D as KPDetector3D_integral
from modules.keypoint_detector_integral_multi import KPDetector3DMulti as KPDetector3DMulti_integral
from modules.util import convert_patch_to_world, triangulation
from eval_utils import cal_per_class_error
from xas_amd import ops_eval
from xas_amd.synthetic import synthetic_eval_batch

ACTIONS = ('Directions', 'Discussion', 'Eating', 'Greeting', 'Phoning', 'Posing', 'Purchases', 'Sitting',
           'SittingDown', 'Smoking', 'TakingPhoto', 'Waiting', 'Walking', 'WalkDog', 'WalkTogether')
act = {name: 0.0 for name in ACTIONS}
act_idx_2_name = {i + 2: name for i, name in enumerate(ACTIONS)}              # eval.py:31-36
METRICS_3D = (('mpjpe', 0), ('n-mpjpe', 1), ('p-mpjpe', 2))                   # alignment none / scale / procrustes


def update_dict(record_table, count_table, error, act_names):
    """Per-action accumulation (eval.py:38-42): the action id sits in characters 4-5 of the sample's `act`."""
    for i, item in enumerate(act_names):
        name = act_idx_2_name[int(item[4:6])]
        record_table[name] += error[i]
        count_table[name] += 1


def ddp_setup():
    init_process_group(backend='nccl')            # RCCL on ROCm
    torch.cuda.set_device(int(os.environ['LOCAL_RANK']))


class Eval:
    """Same constructor and method signatures as the reference Eval (eval.py:63-107)."""

    def __init__(self, config, detector, eval_data, log_dir, img_size=256.0):
        self.gpu_id = int(os.environ.get('LOCAL_RANK', 0))
        self.config = config
        self.cam_id_list = config['model_params']['cam_id_list']
        self.cal_per_act = config['dataset_params']['dataset']['name'] != 'mpi_inf_3dhp'
        self.detector = detector.to(self.gpu_id)
        self.detector.eval()                       # BN uses running statistics; no DDP wrapper is needed to evaluate
        self.eval_data = eval_data
        self.log_dir = log_dir
        self.img_size = img_size

    def convert_data_to_device(self, x):
        for key, v in x.items():
            if isinstance(v, torch.Tensor):
                x[key] = v.to(self.gpu_id, non_blocking=True)
            elif isinstance(v, dict):
                x[key] = self.convert_data_to_device(v)
            elif isinstance(v, np.ndarray):
                x[key] = torch.from_numpy(v).to(self.gpu_id)
        return x

    @torch.no_grad()
    def eval_batch(self, x, mode='best', kps_by_cam=None):
        """Device part of one iteration of eval.py:111-204.  Returns ({name: tensor on the device}, layout) where
        every tensor is per sample [B] (errors), or a scalar (ambiguity, pck, auc)."""
        cams = ['cam_{}'.format(c) for c in self.cam_id_list]
        sel, out, trans = {}, {}, None
        for m in cams:
            kps = kps_by_cam[m] if kps_by_cam is not None else self.detector(x[m + '_img'])[0]
            s = ops_eval.eval_select(kps, x[m + '_joints'], image_size=self.img_size, mode=mode)
            sel[m] = s['sel3d']
            out['err2d_' + m] = s['err2d']
            t = s['swapped'].float()
            trans = t if trans is None else trans + t
        out['ambiguity'] = torch.minimum(trans, len(cams) - trans).mean()                  # eval.py:164-167
        gt = convert_patch_to_world(x['cam_0_joints'], x, 'cam_0', is_norm=False)          # eval.py:170
        preds = {'tri': triangulation(sel, x, self.cam_id_list)}
        for m in cams:
            preds['view_' + m] = convert_patch_to_world(sel[m], x, m, is_norm=True)
        for name, p in preds.items():
            e = pose_errors(p, gt, want=('err',))['err'].mean(dim=2)                       # [3,B]
            for metric, a in METRICS_3D:
                out['{}_{}'.format(metric, name)] = e[a]
            if not self.cal_per_act:                                                       # eval.py:51-55
                q = pose_errors(p, gt, in_div=1000.0, want=('pck', 'auc_hits'))
                out['pck_' + name] = q['pck'].mean()
                out['auc_' + name] = (q['auc_hits'].sum(dim=0).float() / q['pck'].numel()).mean() * 100
        out['world_gt'], out['world_tri'] = gt, preds['tri']
        return out

    def eval(self, tb_log, record_table, count_table, record_3d_table, count_3d_table, record_3d_tri_table,
             count_3d_tri_table, ambiguity_ratio, mode='best'):
        cams = ['cam_{}'.format(c) for c in self.cam_id_list]
        for x in self.eval_data:
            x = self.convert_data_to_device(x)
            out = self.eval_batch(x, mode)
            host = {k: v.detach().cpu().numpy() for k, v in out.items() if not k.startswith('world')}   # the one sync
            for m in cams:
                if self.cal_per_act:
                    update_dict(record_table, count_table, host['err2d_' + m], x['act'])
                else:
                    record_table = record_table + host['err2d_' + m]
                    count_table += 1
            ambiguity_ratio = ambiguity_ratio + float(host['ambiguity'])
            for table, count, names in ((record_3d_tri_table, count_3d_tri_table, ['tri']),
                                        (record_3d_table, count_3d_table, ['view_' + m for m in cams])):
                for name in names:
                    for metric, _ in METRICS_3D:
                        err = host['{}_{}'.format(metric, name)]
                        if self.cal_per_act:
                            update_dict(table[metric], count[metric], err, x['act'])
                        else:
                            table[metric] = table[metric] + err
                            count[metric] += 1
                    if not self.cal_per_act:
                        for key in ('pck', 'auc'):
                            table[key] += float(host['{}_{}'.format(key, name)])
                            count[key] += 1
        return [record_table, count_table, record_3d_table, count_3d_table, record_3d_tri_table, count_3d_tri_table,
                ambiguity_ratio]

    def record(self, record_table, count_table, record_3d_table, count_3d_table, record_3d_tri_table,
               count_3d_tri_table, ambiguity_ratio):
        """Prints and writes <log_dir>/eval/eval_result.txt with the reference's lines (eval.py:206-296)."""
        lines = []
        if self.cal_per_act:
            full2d, sel2d = cal_per_class_error(record_table, count_table)
            full3d, sel3d = cal_per_class_error(record_3d_table, count_3d_table, multi=True)
            fulltri, seltri = cal_per_class_error(record_3d_tri_table, count_3d_tri_table, multi=True)
            for title, e2, e3, et in (('', full2d, full3d, fulltri), ('--------select---------', sel2d, sel3d, seltri)):
                if title:
                    lines.append(title)
                lines.append('2D MSE: {} %'.format(float(e2)))
                for tag, tbl in (('', e3), ('TRI ', et)):
                    for label, key in (('MPJPE', 'mpjpe'), ('N-MPJPE', 'n-mpjpe'), ('P-MPJPE', 'p-mpjpe')):
                        lines.append('{}{}: {} %'.format(tag, label, float(tbl[key])))
        else:
            lines.append('2D MSE: {} %'.format(float(np.mean(record_table) / count_table)))
            for title, tbl, cnt in (('---3D-----', record_3d_table, count_3d_table),
                                    ('---Tri3D-----', record_3d_tri_table, count_3d_tri_table)):
                lines.append(title)
                for key in tbl.keys():
                    if key in ('pck', 'auc'):
                        lines.append('{}: {} %'.format(key, float(tbl[key] / cnt[key])))
                    else:
                        lines.append('{}: {}'.format(key, float(np.mean(tbl[key]) / cnt[key])))
        print('\n'.join(lines))
        os.makedirs(os.path.join(self.log_dir, 'eval'), exist_ok=True)
        path = os.path.join(self.log_dir, 'eval', 'eval_result.txt')
        with open(path, 'w') as f:
            f.write('\n'.join(lines) + '\n')
        print('Results saved in {}'.format(path))
        print('Ambiguity Ratio:{}'.format(ambiguity_ratio / len(self.eval_data) / len(self.cam_id_list)))
        return lines


def prepare_model(config, opt):
    """Detector from the `unsup_model` entry of a training checkpoint (eval.py:298-314)."""
    dp = config['model_params']['detector_params']
    detector = (KPDetector3DMulti_integral if dp['name'] == 'resnet_multi' else KPDetector3D_integral)(**dp)
    if opt.checkpoint is not None:
        checkpoint = torch.load(opt.checkpoint, map_location='cpu')
        detector.load_state_dict({k.replace('regressor.', ''): v for k, v in checkpoint['unsup_model'].items()
                                  if 'regressor.' in k})
    return detector


class _SyntheticEvalLoader:
    def __init__(self, steps, batch, cams, device, seed=0):
        self.steps, self.batch, self.cams, self.device, self.seed = steps, batch, cams, device, seed

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield synthetic_eval_batch(self.batch, self.cams, self.device, seed=self.seed + i)


def prepare_data(config, world_size, worker, synthetic_steps=0, device='cuda'):
    bs = config['train_params']['batch_size'] // world_size
    if synthetic_steps:
        return _SyntheticEvalLoader(synthetic_steps, bs, config['dataset_params']['cam_id_list'], device)
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from train_util import basic_data              # the reference's CPU dataset code (cv2, scikit-fmm)
    ds = basic_data(config, eval_only=True)
    return DataLoader(ds, batch_size=bs, shuffle=False, num_workers=worker, drop_last=False, pin_memory=True,
                      sampler=DistributedSampler(ds, shuffle=False))


def init_tables(cal_per_act):
    """The seven accumulators of eval.py:344-376."""
    if cal_per_act:
        three = lambda: {m: copy.deepcopy(act) for m, _ in METRICS_3D}
        return [copy.deepcopy(act), copy.deepcopy(act), three(), three(), three(), three(), 0.0]
    flat = lambda: {'mpjpe': 0.0, 'n-mpjpe': 0.0, 'p-mpjpe': 0.0, 'pck': 0.0, 'auc': 0.0}
    return [0.0, 0.0, flat(), flat(), flat(), flat(), 0.0]


CLI = (  # same flags as the reference entry (eval.py:326-334) + --synthetic
    ('--config', dict(required=True, help='path to config')),
    ('--log_dir', dict(default='log', help='path to log into')),
    ('--checkpoint', dict(default=None, help='path to checkpoint to restore')),
    ('--batch_size', dict(default=None, type=int)),
    ('--worker', dict(default=10, type=int)),
    ('--extra_tag', dict(default=' ')),
    ('--multi_hypo', dict(default='best', choices=['best', 'confident'], help='multi-hypothesis eval mode')),
    ('--synthetic', dict(default=0, type=int, help='evaluate N synthetic batches (no dataset on this machine)')),
)


def main():
    parser = ArgumentParser()
    for flag, kw in CLI:
        parser.add_argument(flag, **kw)
    opt = parser.parse_args()
    with open(opt.config) as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    config['model_params']['cam_id_list'] = config['dataset_params']['cam_id_list']
    if opt.batch_size:
        config['train_params']['batch_size'] = opt.batch_size
    if opt.checkpoint is None and not opt.synthetic:
        raise Exception('Must specify checkpoint path')
    ddp_setup()
    rank_local, world = int(os.environ['LOCAL_RANK']), int(os.environ['WORLD_SIZE'])
    log_dir = os.path.dirname(opt.checkpoint) if opt.checkpoint else opt.log_dir
    detector = prepare_model(config, opt)
    loader = prepare_data(config, world, opt.worker, opt.synthetic, torch.device('cuda', rank_local))
    ev = Eval(config, detector, loader, log_dir)
    tables = init_tables(ev.cal_per_act)
    record = ev.eval(None, *tables, mode=opt.multi_hypo)
    destroy_process_group()
    if rank_local == 0:
        ev.record(*record)


if __name__ == '__main__':
    main()
