#!/usr/bin/env python3
"""torchrun entry of the MI355X-native trainer (reference: train.py:1-344, same CLI and checkpoint format).

    torchrun --nproc-per-node=N train.py --config config/HM36_Multi_SurS1.yaml [--checkpoint ... --finetune]
             [--batch_size B --epoch E --worker W --seed S --extra_tag T --log_dir D] [--synthetic STEPS]

Differences from the reference, all outside the hot step: the two DistributedDataParallel wrappers are
replaced by xas_amd.engine.TrainStep (flat gradient arenas + bucketed RCCL all-reduce on a side stream);
TensorBoard logging is optional (skipped when tensorboard is not installed); `--synthetic STEPS` trains on
synthetic batches with the dataloader's key contract when the licensed datasets / cv2 / scikit-fmm loader
(`train_util.basic_data`, taken from the reference checkout on PYTHONPATH) are unavailable.
"""
import os
import random
from argparse import ArgumentParser
from time import gmtime, strftime

import numpy as np

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (RCCL between processes) on this driver
import torch
import yaml
from torch.distributed import destroy_process_group, init_process_group
from torch.optim.lr_scheduler import MultiStepLR

from train_util import tb_vis
from xas_amd import engine
from xas_amd.synthetic import synthetic_batch


def setup_seed(seed):
    if seed != -1:
        torch.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
        np.random.seed(seed)
        random.seed(seed)


def ddp_setup():
    init_process_group(backend='nccl')            # RCCL on ROCm
    torch.cuda.set_device(int(os.environ['LOCAL_RANK']))


class Trainer:
    """Same constructor arguments as the reference Trainer (train.py:47-59)."""

    def __init__(self, config, unsup_model, unsup_disc, train_data, optimizer_detector, save_dir,
                 checkpoint_path=None, optimizer_discriminator=None, mode='train'):
        self.gpu_id = int(os.environ['LOCAL_RANK'])
        self.unsup_model = unsup_model.to(self.gpu_id)
        self.unsup_disc = unsup_disc.to(self.gpu_id)
        self.train_data = train_data
        self.optimizer_detector = optimizer_detector
        self.optimizer_discriminator = optimizer_discriminator
        self.epochs_run = 0
        self.config = config
        self.tb_parent_ids = np.array(config['model_params']['parent_ids'])      # train.py:91-92
        self.tb_pair_ids = np.array(config['model_params']['flip_pairs'])
        self.save_dir = save_dir
        if checkpoint_path is not None:
            self._load_checkpoint(checkpoint_path, mode)
        tp = config['train_params']
        self.scheduler_detector = MultiStepLR(optimizer_detector, tp['epoch_milestones'], gamma=0.1,
                                              last_epoch=-1 + self.epochs_run * (tp['lr_kp_detector'] != 0))
        self.scheduler_discriminator = None
        if optimizer_discriminator is not None:
            self.scheduler_discriminator = MultiStepLR(optimizer_discriminator, tp['epoch_milestones'], gamma=0.1,
                                                       last_epoch=-1 + self.epochs_run * (tp['lr_discriminator'] != 0))
        self.step = engine.TrainStep(config, self.unsup_model, self.unsup_disc, optimizer_detector,
                                     optimizer_discriminator)

    def _load_checkpoint(self, checkpoint_path, mode):
        ckpt = torch.load(checkpoint_path, map_location=f'cuda:{self.gpu_id}')
        self.unsup_model.load_state_dict(ckpt['unsup_model'], strict=False)
        self.optimizer_detector.load_state_dict(ckpt['optimizer_detector'])
        try:
            self.unsup_disc.load_state_dict(ckpt['unsup_disc'], strict=False)
            if self.optimizer_discriminator is not None:
                self.optimizer_discriminator.load_state_dict(ckpt['optimizer_discriminator'])
        except Exception:
            print('Load new discriminator for ablation')
        if mode == 'train':
            self.epochs_run = ckpt['epochs']
            print(f'Resuming training from checkpoint at Epoch {self.epochs_run}')
        elif mode == 'finetune':
            print(f'Finetuning from checkpoint at Epoch {self.epochs_run}')
        else:
            raise NotImplementedError

    def _save_checkpoint(self, epoch):
        ckpt = {'unsup_model': self.unsup_model.state_dict(), 'unsup_disc': self.unsup_disc.state_dict(),
                'epochs': epoch, 'optimizer_detector': self.optimizer_detector.state_dict(),
                'optimizer_discriminator': self.optimizer_discriminator.state_dict()}
        torch.save(ckpt, os.path.join(self.save_dir, '{:05d}_ckpt.pth.tar'.format(epoch)))

    def convert_data_to_device(self, x):
        for key in x:
            if isinstance(x[key], torch.Tensor):
                x[key] = x[key].to(self.gpu_id, non_blocking=True)
            elif isinstance(x[key], dict):
                x[key] = self.convert_data_to_device(x[key])
            elif isinstance(x[key], np.ndarray):
                x[key] = torch.tensor(x[key]).to(self.gpu_id)
        return x

    def train(self, tb_logger=None):
        tp = self.config['train_params']
        for epoch in range(self.epochs_run, tp['num_epochs']):
            if hasattr(self.train_data, 'sampler') and hasattr(self.train_data.sampler, 'set_epoch'):
                self.train_data.sampler.set_epoch(epoch)
            for it, x in enumerate(self.train_data):
                x = self.convert_data_to_device(x)
                loss_disc, loss_kp, total, output = self.step(x)
                if self.gpu_id == 0 and tb_logger is not None:      # train.py:192-199: every step, total = None on steps
                    cur = epoch * len(self.train_data) + it          # where only the discriminator was updated
                    tb_vis(tb_logger, cur, self.tb_pair_ids, self.tb_parent_ids,
                           total.detach().item() if total is not None else None, loss_kp, loss_disc, output, x, self.config,
                           self.scheduler_detector)
            self.scheduler_detector.step()
            if self.scheduler_discriminator is not None:
                self.scheduler_discriminator.step()
            if self.gpu_id == 0 and (epoch % tp['checkpoint_freq'] == 0 or epoch == tp['num_epochs'] - 1):
                self._save_checkpoint(epoch)


class _SyntheticLoader:
    def __init__(self, steps, batch, cams, device, seed):
        self.steps, self.batch, self.cams, self.device, self.seed = steps, batch, cams, device, seed

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield synthetic_batch(self.batch, self.cams, self.device, seed=self.seed + i)


def prepare_data(config, world_size, worker, synthetic_steps, device, seed):
    bs = config['train_params']['batch_size'] // world_size            # batch_size is GLOBAL (train.py:274)
    if synthetic_steps:
        return _SyntheticLoader(synthetic_steps, bs, config['dataset_params']['cam_id_list'], device, max(seed, 0))
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from train_util import basic_data              # the reference's CPU dataset code (cv2, scikit-fmm)
    ds = basic_data(config)
    return DataLoader(ds, batch_size=bs, shuffle=False, pin_memory=True, num_workers=worker, sampler=DistributedSampler(ds))


def create_logger(opt):
    if opt.checkpoint is not None and not opt.finetune:
        log_dir = os.path.dirname(opt.checkpoint)
    else:
        seed = 'seed{}_'.format(opt.seed if opt.seed != -1 else '_rand')
        log_dir = os.path.join(opt.log_dir, os.path.basename(opt.config).split('.')[0])
        if opt.finetune:
            log_dir += '_FINETUNE'
        log_dir += '_' + seed + opt.extra_tag + strftime('%d_%m_%y_%H.%M.%S', gmtime())
    tb = None
    if os.environ['LOCAL_RANK'] == '0':
        os.makedirs(log_dir, exist_ok=True)
        try:
            from torch.utils.tensorboard import SummaryWriter
            tb = SummaryWriter(log_dir=os.path.join(log_dir, 'tensorboard'))
        except Exception:                      # TensorBoard not installed: same calls, plain files
            from train_util import JsonlWriter
            tb = JsonlWriter(os.path.join(log_dir, 'tensorboard'))
    return log_dir, tb


CLI = (  # same flags as the reference entry (train.py:305-315) + --synthetic
    ('--config', dict(required=True, help='path to config')),
    ('--log_dir', dict(default='log', help='path to log into')),
    ('--checkpoint', dict(default=None, help='path to checkpoint to restore')),
    ('--batch_size', dict(default=None, type=int)),
    ('--epoch', dict(default=None, type=int)),
    ('--worker', dict(default=10, type=int)),
    ('--extra_tag', dict(default='')),
    ('--finetune', dict(default=False, action='store_true', help='finetune the model')),
    ('--seed', dict(default=-1, type=int)),
    ('--synthetic', dict(default=0, type=int, help='train on N synthetic batches per epoch')),
)


def main():
    parser = ArgumentParser()
    for flag, kw in CLI:
        parser.add_argument(flag, **kw)
    opt = parser.parse_args()
    with open(opt.config) as f:
        config = yaml.load(f, Loader=yaml.FullLoader)
    config['model_params']['cam_id_list'] = config['dataset_params']['cam_id_list']
    for key, val in (('batch_size', opt.batch_size), ('num_epochs', opt.epoch)):
        if val:
            config['train_params'][key] = val
    ddp_setup()
    setup_seed(opt.seed)
    rank_local, world = int(os.environ['LOCAL_RANK']), int(os.environ['WORLD_SIZE'])
    save_dir, tb_logger = create_logger(opt)
    nets = engine.prepare_model(config)
    loader = prepare_data(config, world, opt.worker, opt.synthetic, torch.device('cuda', rank_local), opt.seed)
    trainer = Trainer(config, nets[0], nets[1], loader, nets[2], save_dir, checkpoint_path=opt.checkpoint,
                      optimizer_discriminator=nets[3], mode='finetune' if opt.finetune else 'train')
    trainer.train(tb_logger)
    if tb_logger is not None:
        tb_logger.close()
    destroy_process_group()


if __name__ == '__main__':
    main()
