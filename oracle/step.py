"""Oracle: generator / discriminator forward graphs and one optimisation step.
TEST INFRASTRUCTURE ONLY.

Follows modules/model.py:50-192 (Counter3DModel.forward), modules/model.py:218-264
(Counter3DDisc.forward) and the step body at train.py:154-190.
Visualisation-only outputs (the second return value of the reference forwards)
are not produced, except what tests compare.
"""
import torch
import torch.nn as nn

from . import geometry as geo
from . import head as ohead
from . import losses as L
from .nets import Detector


class Regressor(nn.Module):
    """KPDetector3DMulti (name == 'resnet_multi') or KPDetector3D."""

    def __init__(self, name, num_kp, depth_dim, num_hypo=1, neighbor_size=1, num_layers=50):
        super().__init__()
        self.name, self.num_kp = name, num_kp
        self.num_hypo, self.neighbor_size = num_hypo, neighbor_size
        self.multi = name == 'resnet_multi'
        self.net = Detector(num_kp, depth_dim, num_layers).net

    def forward(self, img):
        logits = self.net(img)
        if self.multi:
            kps, dmap, _ = ohead.softargmax_multi(logits, self.num_kp, self.num_hypo, self.neighbor_size)
        else:
            kps, dmap = ohead.softargmax_single(logits, self.num_kp)
        return kps, dmap


def _cam_args(x, key):
    return (x[key + '_trans_image'], x[key + '_k_mat'], x[key + '_pelvis'],
            x[key + '_rot_world'], x[key + '_trans_world'])


def generator_losses(cfg, regressor, physique, disc, x, return_aux=False):
    """modules/model.py:50-192 for multi-camera batches (no 'cam_mono_img')."""
    lc = cfg['loss_config']
    parents, children = geo.skeleton_links(cfg['parent_ids'], cfg.get('line_select_ids'), False, True)
    width = float(cfg.get('body_width', 3.0)) * 1e-3
    sup_dim = cfg['smpl_disc_params'].get('disc_sup_dim', 3)
    cams = ['cam_%s' % c for c in cfg['cam_id_list']]
    kps, world, recon, aux = {}, {}, {}, {}
    for key in cams:
        img = x[key + '_img']
        kps[key], aux['depth_map_' + key] = regressor(img)
        world[key] = torch.stack([geo.patch_to_world(kps[key][:, h], *_cam_args(x, key), image_size=img.shape[-1])
                                  for h in range(kps[key].shape[1])], dim=1)
        recon[key] = geo.draw_lines_max(kps[key][:, 0, :, :2], img.shape[-1], parents, children, width)
    out = {}
    if 'symmetry_loss' in lc:
        w = lc['symmetry_loss']['weight']
        tot = 0
        for key in cams:
            per_h = []
            for h in range(world[key].shape[1]):
                v = L.bone_sym(world[key][:, h]) * w['bone'] + L.kp_sym(world[key][:, h]) * w['kp']
                if 'kp_2d' in w:
                    v = v + L.kp_sym(kps[key][:, h, :, :2], is_3d=False) * 1e2 * w['kp_2d']
                per_h.append(v)
            tot = tot + torch.stack(per_h).min()
        out['symmetry'] = tot
    if 'smpl_gen_loss' in lc:
        tot = 0
        for key in cams:
            pj = world[key]
            pj = (pj - pj[:, [0]]) / 1000          # model.py:124 indexes the HYPOTHESIS axis
            logits = torch.stack([disc(pj[:, h, :, :sup_dim].detach()) for h in range(pj.shape[1])], dim=1)
            if not cfg['smpl_disc_params'].get('use_aug', False):
                tot = tot + L.disc_loss(logits, None)
            else:           # model.py:132-140: the rotated branch is NOT detached
                rot = torch.stack([disc(geo.random_rotation_3d(pj[:, h])[..., :sup_dim]) for h in range(pj.shape[1])], dim=1)
                tot = tot + L.disc_loss(logits, None) * 0.7 + L.disc_loss(rot, None) * 0.3
        out['smpl_gen'] = tot * lc['smpl_gen_loss']['weight']
    if 'smpl_pseudo_img_loss' in lc:
        tot = 0
        for key in cams:
            pred, _ = regressor(x[key + '_pseudo_img'])
            gt = x[key + '_pseudo_joints']
            tot = tot + torch.stack([L.supervision(pred[:, h], gt) for h in range(pred.shape[1])]).min()
        out['smpl_pseudo_img'] = tot * lc['smpl_pseudo_img_loss']['weight']
    if 'physique_recons_loss' in lc and physique is not None:
        use_w = lc['physique_recons_loss']['use_dis_map']
        tot = 0
        for key in cams:
            m = physique(recon[key])
            aux['mask_physique_' + key] = m
            tot = tot + L.mask_recon(m, x[key + '_mask'], x[key + '_geodesic_dis'] if use_w else None)
        out['physique_recons'] = tot * lc['physique_recons_loss']['weight']
    if 'recons_loss' in lc:
        use_w = lc['recons_loss']['use_dis_map']
        tot = 0
        for key in cams:
            tot = tot + L.mask_recon(recon[key], x[key + '_mask'],
                                     x[key + '_geodesic_dis'] if use_w else None, use_clip=True)
        out['reconstruction'] = tot * lc['recons_loss']['weight']
    if return_aux:
        aux.update({'kps': kps, 'world': world, 'recon': recon})
        return out, aux
    return out


def discriminator_loss(cfg, regressor, disc, x):
    """modules/model.py:218-264: detector forward (train-mode BN, graph
    built, output detached), LSGAN on hypotheses vs pseudo joints in PATCH coordinates."""
    sup_dim = cfg['smpl_disc_params'].get('disc_sup_dim', 3)
    tot = 0
    for c in cfg['cam_id_list']:
        key = 'cam_%s' % c
        pred, _ = regressor(x[key + '_img'])
        fake = torch.stack([disc(pred[:, h, :, :sup_dim].detach()) for h in range(pred.shape[1])], dim=1)
        real = disc(x[key + '_pseudo_joints'][..., :sup_dim])
        if not cfg['smpl_disc_params'].get('use_aug', False):
            tot = tot + L.disc_loss(fake, real)
        else:               # model.py:249-258
            real_world = geo.patch_to_world(x[key + '_pseudo_joints'], *[x[key + s] for s in
                                          ('_trans_image', '_k_mat', '_pelvis', '_rot_world', '_trans_world')],
                                          rect_width=256, mono=True, patch=False)
            rot = geo.random_rotation_3d(real_world)
            tot = tot + L.disc_loss(fake, real) * 0.6 + L.disc_loss(disc(rot[..., :sup_dim]), None) * 0.4
    return tot * cfg['loss_config']['smpl_disc_loss']['weight']


def train_step(cfg, regressor, physique, disc, opt_det, opt_disc, x):
    """train.py:160-190: discriminator update, then generator update."""
    loss_d = discriminator_loss(cfg, regressor, disc, x).mean()
    loss_d.backward()
    opt_disc.step()
    opt_disc.zero_grad()
    losses = generator_losses(cfg, regressor, physique, disc, x)
    loss_g = sum(v.mean() for v in losses.values())
    loss_g.backward()
    opt_det.step()
    opt_det.zero_grad()
    return loss_d.detach(), {k: v.mean().detach() for k, v in losses.items()}
