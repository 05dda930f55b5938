"""Synthetic batch with the dataloader's key contract (human_utils/dataloader/dataloader.py:166-191,221,228),
generated directly in HBM.  Used by bench.py and the entry-point smoke run; there is no dataset
on the benchmark box."""
import math

import torch


def _rotations(B, gen, device):
    q = torch.randn(B, 4, generator=gen, device=device)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).view(B, 3, 3)


def _blob(B, S, gen, device):
    """Union of discs along a random poly-line: a body-like binary mask [B,1,S,S]."""
    ys, xs = torch.meshgrid(torch.arange(S, device=device), torch.arange(S, device=device), indexing='ij')
    pts = torch.rand(B, 6, 2, generator=gen, device=device) * 0.5 * S + 0.25 * S
    m = torch.zeros(B, 1, S, S, device=device)
    for i in range(5):
        for t in torch.linspace(0, 1, 8).tolist():
            c = pts[:, i] * (1 - t) + pts[:, i + 1] * t
            d = (xs[None] - c[:, 0, None, None]) ** 2 + (ys[None] - c[:, 1, None, None]) ** 2
            m[:, 0] = torch.maximum(m[:, 0], (d < (0.06 * S) ** 2).float())
    return m


def synthetic_batch(B, cam_ids, device, seed=0, S=256, K=18):
    gen = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=gen, device=device)
    x = {}
    for cam in cam_ids:
        key = 'cam_%s' % cam
        mask = _blob(B, S, gen, device)
        x[key + '_mask'] = mask
        x[key + '_img'] = r(B, 3, S, S) * mask                       # rm_bg: image * mask, /255 normalised
        x[key + '_geodesic_dis'] = 1.0 + 24.0 * r(B, 1, S, S)
        x[key + '_geodesic_center'] = torch.full((B, 1, 2), S // 2, dtype=torch.int16, device=device)   # geodesic.py:19
        x[key + '_img_path'] = ['synthetic/%s/%06d.jpg' % (key, seed * 1000 + i) for i in range(B)]        # dataloader.py:168
        j = 40 + 176 * r(B, K, 3)
        j[..., 2] = 80 * r(B, K) - 40
        x[key + '_joints'] = j
        scale, th = 0.24 + 0.08 * r(B), 0.4 * r(B) - 0.2
        ti = torch.zeros(B, 2, 3, device=device)
        ti[:, 0, 0], ti[:, 0, 1] = scale * torch.cos(th), -scale * torch.sin(th)
        ti[:, 1, 0], ti[:, 1, 1] = scale * torch.sin(th), scale * torch.cos(th)
        ti[:, :, 2] = 80 * r(B, 2) - 40
        x[key + '_trans_image'] = ti
        km = torch.zeros(B, 3, 3, device=device)
        km[:, 0, 0], km[:, 1, 1], km[:, 2, 2] = 1100 + 100 * r(B), 1100 + 100 * r(B), 1.0
        km[:, 0, 2], km[:, 1, 2] = 480 + 60 * r(B), 480 + 60 * r(B)
        x[key + '_k_mat'] = km
        x[key + '_pelvis'] = torch.stack([1000 * r(B) - 500, 1000 * r(B) - 500, 4000 + 2000 * r(B)], 1)
        x[key + '_rot_world'] = _rotations(B, gen, device)
        x[key + '_trans_world'] = 6000 * r(B, 3) - 3000
        x[key + '_pseudo_img'] = r(B, 3, S, S) * _blob(B, S, gen, device)
        pj = 1.6 * r(B, K, 3) - 0.8
        pj[..., 2] = 0.8 * r(B, K) - 0.4
        x[key + '_pseudo_joints'] = pj
    x['act'] = ['act_%02d_subact_01' % (2 + (seed + i) % 15) for i in range(B)]                           # dataloader.py:228
    return x


CONFIG_NAMES = ('HM36_Multi_SurS1', 'HM36_Multi_SurS2', 'HM36_Multi_SynthS1', 'HM36_Multi_SynthS2',
                'MPI_Multi_SurS1', 'MPI_Multi_SurS2', 'MPI_Multi_SynthS2')


def model_config(name='HM36_Multi_SurS1'):
    """dataset / model / train parameters of the shipped YAMLs (config/HM36_Multi_SurS1.yaml:1-106 and its six
    siblings, which differ only in the fields set below; asserted equal to the YAML contents in
    tests/test_configs.py).  Real runs load the YAML itself (train.py); benchmarks and tests have no config dir."""
    if name not in CONFIG_NAMES:
        raise KeyError('unknown config %r (known: %s)' % (name, ', '.join(CONFIG_NAMES)))
    mpi, s2, synth = name.startswith('MPI'), name.endswith('S2'), 'Synth' in name
    cams = [0, 2, 4, 7, 8] if mpi else [0, 1, 2, 3]
    lc = {'recons_loss': {'use_dis_map': not s2, 'weight': 0.02 if s2 else 0.0},
          'physique_recons_loss': {'use_dis_map': not s2, 'weight': 0.02 if s2 else 0.0},
          'smpl_pseudo_img_loss': {'weight': 1.0 if (synth or name == 'MPI_Multi_SurS1') else 3.0}}
    if s2:
        sym = 0.05 if mpi else 0.1
        lc['symmetry_loss'] = {'weight': {'bone': sym, 'kp': sym, 'kp_2d': 0.0}}
    adv = (1.0 if mpi else 0.5) if s2 else 0.0
    lc['smpl_disc_loss'] = {'weight': adv, 'update_interval': 1}
    lc['smpl_gen_loss'] = {'weight': adv}
    mp = {'detector_params': {'name': 'resnet_multi', 'num_kp': 18, 'depth_dim': 64, 'num_hypo': 3, 'neighbor_size': 15},
          'smpl_disc_params': {'name': 'res_sage_gcn_decouple', 'input_dim': 128, 'hidden_dim': 128, 'output_dim': 128,
                               'num_node': 18, 'disc_sup_dim': 3, 'num_layers': 2, 'use_self_loop': True, 'use_pe': True},
          'smpl_layer_params': {'model_path': 'data/smpl_models'},
          'physique_mask_generator_params': {'layers': [32, 64, 128]},
          'parent_ids': [0, 0, 1, 2, 0, 4, 5, 0, 17, 8, 9, 17, 11, 12, 17, 14, 15, 7],
          'child_ids': list(range(18)),
          'flip_pairs': [[1, 4], [2, 5], [3, 6], [14, 11], [15, 12], [16, 13]],
          'line_select_ids': list(range(17)), 'body_width': 3.0, 'loss_config': lc, 'cam_id_list': cams}
    if s2:
        epochs = (10 if name == 'MPI_Multi_SurS2' else 15)
    else:
        epochs = 80 if mpi else 50
    tp = {'num_epochs': epochs, 'batch_size': 32, 'epoch_milestones': [70] if name == 'MPI_Multi_SurS1' else [40],
          'lr_kp_detector': 1.0e-4 if s2 else 2.0e-4, 'lr_discriminator': 1.0e-4 if s2 else 2.0e-4,
          'checkpoint_freq': 2 if s2 else 20, 'patch_width': 256, 'patch_height': 256,
          'rect_3d_width': 2000, 'rect_3d_height': 2000,
          'aug': {'scale_factor': 0.0, 'rot_factor': 0, 'color_factor': 0.0, 'rot_aug_rate': 0.0, 'flip_aug_rate': 0.0,
                  'do_flip_aug': False}}
    ds = {'name': 'mpi_inf_3dhp' if mpi else 'hm36'}           # config/*.yaml:3-4 (eval.py:73 reads it)
    return {'dataset_params': {'dataset': ds, 'cam_id_list': cams, 'geodesic_param_list': [2, 1, 3, 20, 0.0]},
            'model_params': mp, 'train_params': tp}


def synthetic_eval_batch(B, cam_ids, device, seed=0, S=256, K=18, rect=2000.0):
    """Evaluation batch with a geometrically CONSISTENT scene: one set of world joints seen by every camera
    (pin-hole projection, crop affine, depth in patch pixels), so that triangulation and the world-space metrics
    of eval.py:169-202 are meaningful.  Same keys as the dataloader contract plus `act` (eval.py:40-41)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=gen, device=device)
    world = 350.0 * torch.randn(B, K, 3, generator=gen, device=device)
    world[:, 0] = 50.0 * torch.randn(B, 3, generator=gen, device=device)
    x = {'act': ['act_%02d_subact_01_ca_01' % (2 + (seed + i) % 15) for i in range(B)]}
    for cam in cam_ids:
        key = 'cam_%s' % cam
        R = _rotations(B, gen, device)
        t = torch.stack([600 * r(B) - 300, 600 * r(B) - 300, 4500 + 1000 * r(B)], 1)
        camp = torch.einsum('bij,bkj->bki', R, world) + t[:, None]
        fx, fy, cx, cy = 1100 + 100 * r(B), 1100 + 100 * r(B), 480 + 60 * r(B), 480 + 60 * r(B)
        uv = torch.stack([camp[..., 0] / camp[..., 2] * fx[:, None] + cx[:, None],
                          camp[..., 1] / camp[..., 2] * fy[:, None] + cy[:, None]], -1)
        scale, th = 0.24 + 0.08 * r(B), 0.4 * r(B) - 0.2
        A = torch.stack([scale * torch.cos(th), -scale * torch.sin(th), scale * torch.sin(th), scale * torch.cos(th)], 1).view(B, 2, 2)
        off = S / 2 + 12 * r(B, 2) - 6 - torch.einsum('bij,bj->bi', A, uv[:, 0])
        ti = torch.cat([A, off[:, :, None]], dim=2)
        joints = torch.cat([torch.einsum('bij,bkj->bki', A, uv) + off[:, None],
                            ((camp[..., 2] - camp[:, :1, 2]) / (rect / S))[..., None]], dim=-1)
        km = torch.zeros(B, 3, 3, device=device)
        km[:, 0, 0], km[:, 1, 1], km[:, 2, 2], km[:, 0, 2], km[:, 1, 2] = fx, fy, 1.0, cx, cy
        mask = _blob(B, S, gen, device)
        x[key + '_mask'] = mask
        x[key + '_img'] = r(B, 3, S, S) * mask
        x[key + '_joints'], x[key + '_trans_image'], x[key + '_k_mat'] = joints, ti, km
        x[key + '_pelvis'], x[key + '_rot_world'], x[key + '_trans_world'] = camp[:, 0].clone(), R, t
    x['world'] = world
    return x
