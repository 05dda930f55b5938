"""Generator-side and discriminator-side training graphs (reference: modules/model.py:8-264).

Same constructor signatures, attributes (`regressor`, `smpl_discriminator`) and output-dict keys as the
reference; inside, each camera does: fused detector -> ONE patch->world launch for all hypotheses ->
ONE fused line-mask launch -> fused mask losses.  Quirks of the reference are reproduced on purpose:
the `smpl_gen` input subtracts HYPOTHESIS 0 (model.py:124 indexes axis 1) and is detached, so the
regressor receives no adversarial gradient; symmetry / pseudo losses take the min over hypotheses of
batch means; zero-weight losses are still computed and back-propagated.
"""
import os

import torch

from modules.base_losses.loss_func import (compute_disc_loss, compute_mask_reconstruction_loss,
                                           compute_supervision_min, compute_symmetry_min)
from modules.util import convert_patch_to_world, draw_lines_max, random_rotation_3D
from xas_amd import ops_nn, streams


def cal_links(parent_ids, line_select_ids=None, use_root=False, extension=True):
    """(parent, child) joint lists of the drawn lines: the selected skeleton bones plus, optionally, eight
    torso links that fill the trunk (model.py:8-22)."""
    pairs = list(zip(parent_ids, range(len(parent_ids))))
    if not use_root:
        pairs = pairs[1:]
    pairs = [pairs[i] for i in line_select_ids]
    if extension:
        pairs += list(zip([7, 7, 7, 7, 0, 0, 1, 4], [1, 4, 11, 14, 2, 5, 14, 11]))
    return [p for p, _ in pairs], [c for _, c in pairs]


def _cams(x, cam_id_list):
    return ['mono'] if 'cam_mono_img' in x else cam_id_list


def _disc_many(disc, inputs):
    """Logits of the discriminator on a list of inputs: one batched pass when the module offers it."""
    if hasattr(disc, 'forward_groups'):
        return disc.forward_groups(inputs)
    return [disc(t) for t in inputs]


def _to_world(kps, x, key, mono):
    if mono:
        return convert_patch_to_world(kps, x, key, is_norm=True, RECT_WIDTH=256, mono=True, patch=False)
    return convert_patch_to_world(kps, x, key, is_norm=True)


# Camera batching (MI355X: 288 GB of HBM): the reference calls the detector once per camera (model.py:64,147,231) and
# the physique net once per camera (model.py:81).  Here the images of all cameras go through the network as ONE tensor
# [G*B, ...]: 4x larger GEMMs for the small late layers, 4x fewer launches and SyncBatchNorm messages.  Every batch-norm
# layer still normalises each camera's B images with their own statistics and applies the running-statistic updates
# in camera order (ops_nn.bn_groups), so the arithmetic is that of the per-camera calls.  XAS_CAM_BATCH=0 restores one
# call per camera; XAS_CAM_BATCH_MAX caps the number of cameras per pass.
CAM_BATCH = os.environ.get('XAS_CAM_BATCH', '1') == '1'
CAM_BATCH_MAX = max(1, int(os.environ.get('XAS_CAM_BATCH_MAX', '8')))
JOIN_PSEUDO = os.environ.get('XAS_CAM_BATCH_PSEUDO', '1') == '1'
# r04: the discriminator step's detector pass (real images, no graph) rides in front of the generator step's pass (real +
# pseudo images) as a no-grad PREFIX of one grouped pass (ops_nn: prefix pass).  Every one of the reference's 3 * cameras
# detector calls is still computed, in the reference's order; only the launches are shared.  XAS_JOINT_DISC=0: a pass of its own.
JOINT_DISC = os.environ.get('XAS_JOINT_DISC', '1') == '1'


def _grouped(net, tensors):
    """[net(t) for t in tensors] (results as a list), camera-batched when the network offers forward_groups."""
    if not (CAM_BATCH and len(tensors) > 1 and hasattr(net, 'forward_groups')
            and all(t.shape == tensors[0].shape for t in tensors)):
        return [net(t) for t in tensors]
    res = []
    B = tensors[0].shape[0]
    per_pass = max(1, min(CAM_BATCH_MAX, 384 // max(1, B)))     # logits of a pass: G*B*64*64*1152 elements < 2^31
    n_pass = -(-len(tensors) // per_pass)
    per_pass = -(-len(tensors) // n_pass)                        # equal passes (8 groups at B = 64: 4 + 4, not 6 + 2)
    for lo in range(0, len(tensors), per_pass):
        part = tensors[lo:lo + per_pass]
        G = len(part)
        y = net.forward_groups(ops_nn.stack_nchw(part) if G > 1 else part[0], G)
        if isinstance(y, tuple):                         # detector: (kps [G*B, ...], depth maps [G, K, D])
            kps, dmap = y
            dmap = dmap.reshape(G, *dmap.shape[-2:])
            res += [(kps[g * B:(g + 1) * B], dmap[g]) for g in range(G)]
        else:
            res += [y[g * B:(g + 1) * B] for g in range(G)]
    return res


class Counter3DModel(torch.nn.Module):
    def __init__(self, cfg, regressor, smpl_layer, h36m_regressor, physique_network=None):
        super().__init__()
        self.regressor = regressor
        self.cam_id_list = cfg['cam_id_list']
        self.body_width = float(cfg.get('body_width', 3.0)) * 1e-3
        self.parent_ids, self.child_ids = cal_links(cfg['parent_ids'], line_select_ids=cfg.get('line_select_ids'),
                                                    use_root=False, extension=True)
        self.loss_config = cfg['loss_config']
        self.use_learned_width = cfg.get('use_learned_width', False)
        self.smpl_layer = smpl_layer
        self.h36m_regressor = h36m_regressor
        self.physique_network = physique_network
        self.DISC_SUP_DIMENSION = cfg['smpl_disc_params'].get('disc_sup_dim', 3)
        self.use_aug = cfg['smpl_disc_params'].get('use_aug', False)

    def forward(self, x, smpl_discriminator):
        return self.finish(x, smpl_discriminator, *self.camera_passes(x))

    def joint_pass_possible(self, x):
        """Can the discriminator step's detector pass join the generator step's as a no-grad prefix?  One stream, camera
        batching with the pseudo images joined, a multi-hypothesis detector in training mode on the GPU, and all 3 * cameras
        groups within the image budget of one pass (the logits stay below 2^31 elements)."""
        keys = ['cam_{}'.format(c) for c in _cams(x, self.cam_id_list)]
        img = x[keys[0] + '_img']
        groups = 3 * len(keys)
        return (JOINT_DISC and CAM_BATCH and JOIN_PSEUDO and CAM_BATCH_MAX >= 8 and streams.CHAINS <= 1 and img.is_cuda
                and 'smpl_pseudo_img_loss' in self.loss_config and self.regressor.training
                and type(self.regressor).__name__ == 'KPDetector3DMulti' and groups * img.shape[0] <= 384
                and all(x[k + '_img'].shape == img.shape and x[k + '_pseudo_img'].shape == img.shape for k in keys))

    def joint_detector_pass(self, x):
        """ONE grouped detector pass for the step: [real images of every camera (the discriminator step's calls, model.py:231:
        no graph) | the same real images | the pseudo images (the generator step's calls, model.py:64,147)], in the order of
        the reference's calls - which fixes the order of the batch-norm running-statistic updates.
        -> ({cam_key: kps, no graph} for Counter3DDisc.forward(preds=...), [(kps, depth map)] * 2 cameras for camera_passes(dets=...))."""
        keys = ['cam_{}'.format(c) for c in _cams(x, self.cam_id_list)]
        ops_nn.prepack(self.regressor)
        real = [x[k + '_img'] for k in keys]
        imgs = real + real + [x[k + '_pseudo_img'] for k in keys]
        B, P, G = real[0].shape[0], len(keys), len(imgs)
        buf = ops_nn.stack_nchw(imgs)
        if not buf.is_contiguous(memory_format=torch.channels_last):
            buf = ops_nn.from_nchw(buf)
        ops_nn.mark_prefix_buffer(buf)
        kps, dmap, kps_prefix = self.regressor.forward_groups(buf[P * B:], G, prefix_groups=P)
        dmap = dmap.reshape(G, *dmap.shape[-2:])
        preds = {k: kps_prefix[g * B:(g + 1) * B] for g, k in enumerate(keys)}
        dets = [(kps[g * B:(g + 1) * B], dmap[P + g]) for g in range(G - P)]
        return preds, dets

    def camera_passes(self, x, pseudo=True, after_geometry=None, dets=None):
        """Detector / geometry / mask part of the step for every camera (everything that does not involve the
        discriminator).  `pseudo=False` leaves the pseudo-image branch to a later `pseudo_passes` call.  `dets`: the detector
        outputs of joint_detector_pass (real + pseudo images of every camera) - the detector is then not run here."""
        cams = _cams(x, self.cam_id_list)
        lc = self.loss_config
        out = {}
        ops_nn.prepack(self.regressor)
        if self.physique_network is not None:
            ops_nn.prepack(self.physique_network)
        keys = ['cam_{}'.format(c) for c in cams]
        want_pseudo = pseudo and 'smpl_pseudo_img_loss' in lc
        # Two chains (xas_amd/streams.py: chains, default): the real-image pass (detector, geometry, mask renderer, physique
        # net, mask losses) and the pseudo-image pass are independent until the losses are summed; each runs on its own
        # stream, forward and backward, so the batch norms (HBM) of one fill the convolution (matrix pipe) time of the other.
        # XAS_CHAINS=1: one stream, and the pseudo images of every camera join the real images' detector pass
        # (XAS_CAM_BATCH_PSEUDO=0: a pass of their own).  Real cameras first either way: the order of the reference's
        # detector calls, which fixes the order of the batch-norm running-statistic updates.
        two_chains = want_pseudo and streams.CHAINS > 1 and CAM_BATCH and x[keys[0] + '_img'].is_cuda
        fuse_pseudo = want_pseudo and JOIN_PSEUDO and CAM_BATCH and not two_chains
        per_cam = {}
        with streams.chains(2 if two_chains else 1) as ch:
            with ch.run(0):
                if dets is None:
                    imgs = [x[k + '_img'] for k in keys] + ([x[k + '_pseudo_img'] for k in keys] if fuse_pseudo else [])
                    dets = _grouped(self.regressor, imgs)
                elif not (fuse_pseudo and len(dets) == 2 * len(keys)):
                    raise RuntimeError('camera_passes(dets=...): expects the joined real + pseudo detector outputs')
                pseudo_dets = dets[len(keys):] if fuse_pseudo else None
                for cam, key, (kps, depth_map) in zip(cams, keys, dets):
                    assert kps.dim() == 4, "use aligned multi-hypothesis settings"
                    # (slices, not the reference's `[[0]]` lists: a list index is an index TENSOR that torch uploads with a
                    # blocking copy - 46 pipeline drains per step; the values are the same)
                    out['pose_2d_pred_{}_ori'.format(key)] = kps[0:1, 0].detach()        # (a view: nothing writes kps in place)
                    out['depth_map_{}'.format(key)] = depth_map
                    world = _to_world(kps, x, key, cam == 'mono')                  # [B, Hy, K, 3], one launch
                    out['pose_3d_depth_{}'.format(key)] = world[:, 0].detach()
                    # multi-hypothesis only changes z, so one mask per camera (hypothesis 0's x, y)
                    recon = draw_lines_max(kps[:, 0, :, :2], x[key + '_img'].shape[-1], self.parent_ids, self.child_ids,
                                           self.body_width)
                    out['mask_heatmap_line_{}'.format(key)] = recon.detach()
                    per_cam[key] = dict(kps=kps, world=world, recon=recon)
                if after_geometry is not None:           # the world joints of every camera exist: engine.TrainStep starts the
                    after_geometry(per_cam)              # adversarial term on its second stream, beside the physique net
                if 'physique_recons_loss' in lc and self.physique_network is not None:
                    use_w = lc['physique_recons_loss']['use_dis_map']
                    for key, phys in zip(keys, _grouped(self.physique_network, [per_cam[k]['recon'] for k in keys])):
                        out['mask_physique_{}'.format(key)] = phys[0:1].detach()
                        per_cam[key]['phys'] = compute_mask_reconstruction_loss(
                            phys, x[key + '_mask'], weight=x[key + '_geodesic_dis'] if use_w else None)
                if 'recons_loss' in lc:
                    use_w = lc['recons_loss']['use_dis_map']
                    for key in keys:
                        per_cam[key]['recons'] = compute_mask_reconstruction_loss(
                            per_cam[key]['recon'], x[key + '_mask'], weight=x[key + '_geodesic_dis'] if use_w else None,
                            use_clip=True)
                if pseudo and not two_chains:
                    self.pseudo_passes(x, per_cam, out, pseudo_dets)
            if two_chains:
                with ch.run(1):
                    self.pseudo_passes(x, per_cam, out, None)
            for d in per_cam.values():                       # consumed by `finish` on the entering stream
                ch.to_main(*d.values())
            ch.to_main(*out.values())
        return per_cam, out

    def pseudo_passes(self, x, per_cam, out, dets=None):
        """Pseudo-image branch (model.py:145-164): detector on the synthetic image of every camera, supervised by its
        joints, min over hypotheses of the batch-mean error."""
        if 'smpl_pseudo_img_loss' not in self.loss_config:
            return
        cams = _cams(x, self.cam_id_list)
        keys = ['cam_{}'.format(c) for c in cams]
        if dets is None:
            dets = _grouped(self.regressor, [x[k + '_pseudo_img'] for k in keys])
        for key, (pred, _) in zip(keys, dets):
            gt = x[key + '_pseudo_joints']
            out['pose_2d_pred_{}_pseudo'.format(key)] = pred[0:1, 0].detach()
            out['pose_3d_pred_{}_pseudo'.format(key)] = _to_world(pred[:, 0].detach(), x, key, True)[0:1]
            out['pose_3d_gt_{}_pseudo'.format(key)] = _to_world(gt, x, key, True)[0:1]
            per_cam[key]['pseudo'] = compute_supervision_min(pred, gt)

    def adversarial_term(self, x, smpl_discriminator, world):
        """smpl_gen_loss (model.py:118-143): the discriminator on the DETACHED, root-relative predicted poses of every camera
        and hypothesis (use_aug: plus on randomly rotated, NOT detached ones)."""
        cams = _cams(x, self.cam_id_list)
        total = 0
        rels = {}
        for cam in cams:
            key = 'cam_{}'.format(cam)
            rels[key] = ((world[key] - world[key][:, 0:1]) / 1000)[..., :self.DISC_SUP_DIMENSION].detach()
        hy = world['cam_{}'.format(cams[0])].shape[1]
        ops_nn.prepack(smpl_discriminator)               # (the update has just rewritten its weights: ONE launch re-splits all of them)
        flat = _disc_many(smpl_discriminator, [rels['cam_{}'.format(c)][:, h] for c in cams for h in range(hy)])
        for ci, cam in enumerate(cams):
            key = 'cam_{}'.format(cam)
            rel = rels[key]
            logits = torch.stack(flat[ci * hy:(ci + 1) * hy], dim=1)
            if not self.use_aug:
                total = total + compute_disc_loss(logits, None)
            else:
                rot = torch.stack([smpl_discriminator(random_rotation_3D((world[key] - world[key][:, 0:1])[:, h] / 1000)
                                                      [..., :self.DISC_SUP_DIMENSION]) for h in range(rel.shape[1])], dim=1)
                total = total + compute_disc_loss(logits, None) * 0.7 + compute_disc_loss(rot, None) * 0.3
        return total * self.loss_config['smpl_gen_loss']['weight']

    def adversarial_on(self, aux, x, smpl_discriminator, world):
        """adversarial_term on stream `aux` (which holds the updated discriminator): forward now, backward replayed there."""
        cur = torch.cuda.current_stream()
        aux.wait_stream(cur)
        for t in world.values():
            t.record_stream(aux)
        with torch.cuda.stream(aux):
            return self.adversarial_term(x, smpl_discriminator, world)

    def finish(self, x, smpl_discriminator, per_cam, out, aux=None, gen_val_early=None, wait_for=None):
        """Losses from the per-camera results (model.py:98-190).  aux: a second stream that already holds the updated
        discriminator (engine.TrainStep): the adversarial term - the discriminator on the DETACHED poses (model.py:128), so
        its forward and backward share nothing with the rest of the graph - runs there, beside the other losses and, in
        the backward pass, beside the physique / detector backward (autograd replays a node on the stream it was recorded on).
        r05: NOT the default any more (engine.ADV_ON_AUX): with the term's graph on a second stream one step in ~12 read a few
        64-byte sectors of a small main-stream tensor stale (DESIGN: the r04 driver failure).  wait_for: the stream that
        carried the discriminator update when the term runs on the CURRENT stream - waited for right before the term."""
        cams = _cams(x, self.cam_id_list)
        lc = self.loss_config
        losses = {}
        kps = {k: v['kps'] for k, v in per_cam.items()}
        world = {k: v['world'] for k, v in per_cam.items()}
        if 'mono' not in cams:
            out['kp_gt_world'] = convert_patch_to_world(x['cam_0_joints'], x, 'cam_0', is_norm=False)[0:1]

        gen_val, cur = gen_val_early, None
        if 'smpl_gen_loss' in lc and aux is not None:
            cur = torch.cuda.current_stream()
            if gen_val is None:
                gen_val = self.adversarial_on(aux, x, smpl_discriminator, world)

        if 'symmetry_loss' in lc:
            w = lc['symmetry_loss']['weight']
            total = 0
            for cam in cams:
                if cam == 'mono':
                    continue
                key = 'cam_{}'.format(cam)
                total = total + compute_symmetry_min(world[key], w['bone'], w['kp'], kps[key], w.get('kp_2d'))
            losses['symmetry'] = total

        if 'smpl_gen_loss' in lc:
            if gen_val is None and wait_for is not None:
                # the discriminator update ran on stream `wait_for`: its Adam step precedes the adversarial term (train.py:160-190)
                torch.cuda.current_stream().wait_stream(wait_for)
            losses['smpl_gen'] = gen_val if gen_val is not None else self.adversarial_term(x, smpl_discriminator, world)

        if 'smpl_pseudo_img_loss' in lc:
            losses['smpl_pseudo_img'] = sum(per_cam['cam_{}'.format(c)]['pseudo'] for c in cams) \
                * lc['smpl_pseudo_img_loss']['weight']
        if 'physique_recons_loss' in lc and self.physique_network is not None:
            losses['physique_recons'] = sum(per_cam['cam_{}'.format(c)]['phys'] for c in cams) \
                * lc['physique_recons_loss']['weight']
        if 'recons_loss' in lc:
            losses['reconstruction'] = sum(per_cam['cam_{}'.format(c)]['recons'] for c in cams) * lc['recons_loss']['weight']
        if cur is not None:
            cur.wait_stream(aux)                       # the adversarial term joins the sum of the losses
            gen_val.record_stream(cur)
        return losses, out


class Counter3DDisc(torch.nn.Module):
    def __init__(self, cfg, smpl_discriminator, smpl_layer, h36m_regressor):
        super().__init__()
        self.smpl_discriminator = smpl_discriminator
        self.cam_id_list = cfg['cam_id_list']
        self.parent_ids, self.child_ids = cal_links(cfg['parent_ids'], line_select_ids=cfg.get('line_select_ids'),
                                                    use_root=False, extension=False)
        self.loss_config = cfg['loss_config']
        if 'GCN' in self.smpl_discriminator.name:
            self.smpl_discriminator.parent_ids = self.parent_ids
            self.smpl_discriminator.child_ids = self.child_ids
        self.smpl_layer = smpl_layer
        self.h36m_regressor = h36m_regressor
        self.DISC_SUP_DIMENSION = cfg['smpl_disc_params'].get('disc_sup_dim', 3)
        self.use_aug = cfg['smpl_disc_params'].get('use_aug', False)

    def detector_pass(self, x, regressor):
        """The detector on the real images of every camera for the discriminator update (model.py:231): -> {cam_key: kps}.
        The reference builds (and discards) an autograd graph here (output detached at model.py:243); only the values and
        the train-mode BN running-statistic updates matter, so no graph is recorded."""
        keys = ['cam_{}'.format(c) for c in _cams(x, self.cam_id_list)]
        ops_nn.prepack(regressor)
        with torch.no_grad():
            dets = _grouped(regressor, [x[k + '_img'] for k in keys])
        return {k: kp for k, (kp, _) in zip(keys, dets)}

    def forward(self, x, regressor, preds=None):
        """`preds` (optional, {cam_key: kps [B,Hy,K,3]}): detector outputs already computed on the same images with
        the same weights (engine.TrainStep(dedupe=True)); the detector is then not run again here."""
        total = 0
        out = {}
        d = self.DISC_SUP_DIMENSION
        cams = _cams(x, self.cam_id_list)
        keys = ['cam_{}'.format(c) for c in cams]
        reals, inputs = {k: x[k + '_pseudo_joints'] for k in keys}, []
        if preds is not None:
            preds = {k: v.detach() for k, v in preds.items()}
        else:
            preds = self.detector_pass(x, regressor)
        ops_nn.prepack(self.smpl_discriminator)
        for cam in cams:
            key = 'cam_{}'.format(cam)
            inputs += [preds[key][:, h, :, :d] for h in range(preds[key].shape[1])] + [reals[key][..., :d]]
        logits = _disc_many(self.smpl_discriminator, inputs)           # every hypothesis / camera in one pass
        per_cam = len(inputs) // len(cams)
        for ci, cam in enumerate(cams):
            key = 'cam_{}'.format(cam)
            real = reals[key]
            real_world = _to_world(real, x, key, True)
            out['pose_smpl_2d_{}'.format(key)] = real[0:1]
            out['pose_smpl_3d_{}'.format(key)] = real_world[0:1].detach()
            mine = logits[ci * per_cam:(ci + 1) * per_cam]
            fake_logits, real_logits = torch.stack(mine[:-1], dim=1), mine[-1]
            out['smpl_logits_{}'.format(key)] = real_logits[0:1]
            out['pred_logits_{}'.format(key)] = fake_logits[0:1, 0]
            if self.use_aug:
                rot = random_rotation_3D(real_world)
                out['pose_smpl_3d_{}_rot'.format(key)] = rot[0:1]
                total = total + compute_disc_loss(fake_logits, real_logits) * 0.6 \
                    + compute_disc_loss(self.smpl_discriminator(rot[..., :d]), None) * 0.4
            else:
                total = total + compute_disc_loss(fake_logits, real_logits)
        return total * self.loss_config['smpl_disc_loss']['weight'], out
