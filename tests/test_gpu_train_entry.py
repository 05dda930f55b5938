"""The mirrored torchrun entry runs one tiny epoch on synthetic batches, writes a checkpoint in the reference's
format, and resumes from it."""
import os
import subprocess
import sys

import pytest
import yaml

pytestmark = [pytest.mark.gpu, pytest.mark.multiproc, pytest.mark.limit(400)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'x-as-supervision_amd')


def test_train_entry_synthetic(tmp_path):
    sys.path.insert(0, PKG)
    from xas_amd.synthetic import model_config
    cfg = model_config('HM36_Multi_SurS2')
    cfg['dataset_params']['cam_id_list'] = [0, 1]
    cfg['train_params'].update(num_epochs=1, batch_size=2, checkpoint_freq=1, epoch_milestones=[40])
    cfg_path = tmp_path / 'HM36_Multi_SurS2.yaml'
    cfg_path.write_text(yaml.safe_dump(cfg))
    env = dict(os.environ, PYTHONPATH=PKG, MASTER_ADDR='127.0.0.1')
    base = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
            '--master-port', '29541', os.path.join(PKG, 'train.py'), '--config', str(cfg_path), '--synthetic', '2',
            '--log_dir', str(tmp_path / 'log'), '--seed', '3']
    from _ranks import release_gpu_memory
    release_gpu_memory()                     # the child needs the card's memory, not this process's cache
    r = subprocess.run(base, env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-3000:]
    runs = os.listdir(tmp_path / 'log')
    assert len(runs) == 1
    ckpt = tmp_path / 'log' / runs[0] / '00000_ckpt.pth.tar'
    assert ckpt.exists()
    import torch
    sd = torch.load(ckpt, map_location='cpu')
    assert set(sd) == {'unsup_model', 'unsup_disc', 'epochs', 'optimizer_detector', 'optimizer_discriminator'}
    assert 'regressor.net.backbone.conv1.weight' in sd['unsup_model']
    assert 'physique_network.encoder.0.0.weight' in sd['unsup_model']
    assert 'smpl_discriminator.joint_gcn.0.gc1.lin_l.weight' in sd['unsup_disc']
    assert set(sd['optimizer_detector']['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
    # resume (finetune mode restarts at epoch 0 in a new directory)
    r = subprocess.run(base + ['--checkpoint', str(ckpt), '--finetune'], env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-3000:]
    # evaluate the checkpoint with the mirrored eval entry (synthetic consistent scene): result file as the reference's
    ev = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
          '--master-port', '29542', os.path.join(PKG, 'eval.py'), '--config', str(cfg_path), '--checkpoint', str(ckpt),
          '--synthetic', '2', '--batch_size', '2', '--multi_hypo', 'best']
    r = subprocess.run(ev, env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stderr[-3000:]
    res = (tmp_path / 'log' / runs[0] / 'eval' / 'eval_result.txt').read_text().split('\n')
    assert res[0].startswith('2D MSE: ') and any(l.startswith('TRI P-MPJPE: ') for l in res)
    assert '--------select---------' in res
    assert 'Ambiguity Ratio' in r.stdout
