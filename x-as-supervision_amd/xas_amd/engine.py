"""Model construction and the optimisation step (reference: train.py:147-269).

`prepare_model(config)` mirrors train.py:212-269 (same classes, same optimizer hyper-parameters) with
FusedAdam; `TrainStep` runs exactly the body of the reference's hot loop (train.py:160-190): discriminator
forward/backward/update, then generator forward/backward/update, with data-parallel gradient averaging on a
side stream when a process group is active.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from modules.discriminator import GCNDiscriminator, GCNDiscriminatorDecouple, GCNSAGEDiscriminator
from modules.keypoint_detector_integral import KPDetector3D
from modules.keypoint_detector_integral_multi import KPDetector3DMulti
from modules.model import Counter3DDisc, Counter3DModel
from modules.physique_network import PhysiqueMaskGenerator

from . import ops_nn
from ._lib import call, ptr
from .dp import GradReducer, dp_active, sync_buffers
from .optim import FusedAdam

# The discriminator's own forward / backward / Adam (a few hundred small launches, ~4 ms of an otherwise idle GPU) CAN run on a
# second stream beside the generator step's detector passes, which do not depend on it (only the generator LOSSES do):
# XAS_DISC_BESIDE_GEN=1.  r03 / r04 shipped that as the default (-5.7 ms per step); r05: OFF - see ops_nn._side: with the
# update on a second stream AND the weight gradients on a third, one step in six computed wrong values somewhere (that is what
# the r04 driver run's red test was), and no two-stream combination was clean over 400 steps either.  Under data parallelism
# the update stays on the main stream as well.  tests/test_gpu_dp_step.py runs the two-rank step in both settings.
_BESIDE_ENV = os.environ.get('XAS_DISC_BESIDE_GEN')
DISC_BESIDE_GEN = _BESIDE_ENV is not None and _BESIDE_ENV != '0'          # (kept for tools that read it)
# The adversarial term of the generator step (forward + backward of the discriminator on the detached poses) on the second
# stream too: OFF since r05.  With it on, one step in ~12 computed a different result - a main-stream kernel read three or six
# consecutive 64-byte sectors of a small tensor (joints, world joints) as they had been BEFORE the kernel in front of it on the
# same stream wrote them (tools/diag_repro.py, profiles/r05_step_reproducibility.md); 0 of 200 steps with the term on the
# main stream, 0 of 100 with the whole update there.  XAS_ADV_AUX=1 restores the r04 schedule.
ADV_ON_AUX = os.environ.get('XAS_ADV_AUX', '0') == '1'
_aux = {}
_DEBUG_SYNC = os.environ.get('XAS_DEBUG_SYNC', '').split(',')      # diagnostic sync points (tools/diag_repro.py)


def disc_beside_gen():
    return _BESIDE_ENV is not None and _BESIDE_ENV != '0'


def _aux_stream():
    if os.environ.get('XAS_AUX_IS_SIDE', '0') == '1':       # (diagnostic: ONE secondary stream for the update and the weight gradients)
        return ops_nn.side_stream()
    dev = torch.cuda.current_device()
    if dev not in _aux:
        _aux[dev] = torch.cuda.Stream()
    return _aux[dev]


def prepare_model(config, smpl_arrays=None):
    """-> unsup_model, unsup_disc, optimizer_detector, optimizer_discriminator  (train.py:212-269)."""
    mp, tp = config['model_params'], config['train_params']
    det = mp['detector_params']
    regressor = KPDetector3DMulti(**det) if det['name'] == 'resnet_multi' else KPDetector3D(**det)
    disc = smpl_layer = h36m = None
    if 'smpl_disc_params' in mp:
        name = mp['smpl_disc_params']['name']
        if 'gcn' not in name:
            raise NotImplementedError
        if 'decouple' in name:
            disc = GCNDiscriminatorDecouple(mp['smpl_disc_params'])
        elif 'sage' in name:
            disc = GCNSAGEDiscriminator(mp['smpl_disc_params'])
        else:
            disc = GCNDiscriminator(mp['smpl_disc_params'])
        smpl_layer, h36m = _load_smpl(mp, smpl_arrays)
    phys = None
    if 'physique_mask_generator_params' in mp:
        phys = PhysiqueMaskGenerator(mp['physique_mask_generator_params']['layers'])
    net_params = list(regressor.parameters()) + (list(phys.parameters()) if phys is not None else [])
    opt_det = FusedAdam(net_params, lr=tp['lr_kp_detector'], betas=(0.5, 0.999))
    opt_disc = FusedAdam(disc.parameters(), lr=tp['lr_discriminator'], betas=(0.5, 0.999)) if disc is not None else None
    unsup_model = Counter3DModel(mp, regressor, smpl_layer, h36m, phys)
    unsup_disc = Counter3DDisc(mp, disc, smpl_layer, h36m)
    return unsup_model, unsup_disc, opt_det, opt_disc


def _load_smpl(mp, smpl_arrays):
    """SMPL layer + H36M joint regressor (train.py:230-238).  The layer is constructed for API parity; it is
    not called on the training path.  Without the licensed files (benchmarks, tests) it is skipped."""
    from modules.smplpytorch.pytorch.smpl_layer import SMPL_Layer
    if smpl_arrays is not None:
        return SMPL_Layer.from_arrays(smpl_arrays, center_idx=0), torch.as_tensor(smpl_arrays['h36m_regressor'])
    root = mp.get('smpl_layer_params', {}).get('model_path', '')
    reg = os.path.join(root, 'J_regressor_h36m.npy')
    if os.path.exists(reg):
        return (SMPL_Layer(center_idx=0, gender='neutral', model_root=root),
                torch.tensor(np.load(reg), dtype=torch.float32))
    return None, None


class TrainStep:
    """One optimisation step = train.py:160-190.  Holds the two reducers (DDP equivalents)."""

    def __init__(self, config, unsup_model, unsup_disc, opt_det, opt_disc, num_buckets=4, dedupe=False):
        # dedupe: the reference runs the detector on the real images twice per step with IDENTICAL weights (once
        # detached for the discriminator update, model.py:231, once for the generator losses, model.py:64).  With
        # dedupe=True that forward is computed once; the batch-norm running-statistic updates of the skipped pass
        # are replayed, so parameters AND buffers after the step are bit-identical (tests/test_gpu_model.py).
        # Off by default: the benchmark runs the reference's 12 detector forwards per sample.
        self.dedupe = dedupe
        self.model, self.disc = unsup_model, unsup_disc
        self.opt_det, self.opt_disc = opt_det, opt_disc
        interval = config['model_params']['loss_config']['smpl_disc_loss']['update_interval']
        self.disc_every = interval if interval >= 1 else 1
        self.gen_every = 1 if interval >= 1 else int(1.0 / interval)
        self.cur_step = 0
        self.grad_probe = None      # measurement hook: called as grad_probe('disc' | 'det', gradient arena) just before the
                                    # optimizer consumes it (bench.py's variant check); None on the training path
        self.red_det = self.red_disc = None
        self._weights_checked = False
        self._range_polls = []      # [(event, pinned int32)]: xas_f16_weight_overflow_peek results in flight (read one step late)
        opt_det.grad_arena                       # materialise the gradient arenas: the kernels accumulate weight, bias and
        if opt_disc is not None:                 # norm-parameter gradients straight into them
            opt_disc.grad_arena
        if dp_active():
            f = opt_det
            f.grad_arena
            self.red_det = GradReducer(f._flat['g'], f._flat['params'], f._flat['offs'], num_buckets, name='detector')
            if opt_disc is not None:
                opt_disc.grad_arena
                g = opt_disc
                self.red_disc = GradReducer(g._flat['g'], g._flat['params'], g._flat['offs'], 1, name='discriminator')
            sync_buffers(self.model)
            dist.broadcast(opt_det.param_arena, src=0)
            if opt_disc is not None:
                dist.broadcast(opt_disc.param_arena, src=0)

    def _check_weight_range(self):
        """The f16x3 weight format holds |w| < 64 (2^10 w in fp16): the weight-preparation kernels and the stem kernel flag
        anything larger instead of letting a layer turn into inf / NaN unnoticed.  Tensor operands need no such check: their
        scales come from recorded maxima (ops_nn).  Polled before the FIRST optimizer step - a checkpoint that does not fit
        must not be trained on - and every 64 steps after; one device synchronisation each."""
        if not next(self.model.regressor.parameters()).is_cuda:
            return
        if ops_nn.query('xas_get_precision') == 3 and ops_nn.query('xas_f16_weight_overflow', 1) == 1:
            raise RuntimeError('a convolution weight has left the range of the f16x3 arithmetic (|w| >= 64 or NaN): '
                               'run with XAS_PRECISION=2 (bf16x6)')

    def _poll_weight_range(self):
        """Every step, without a synchronisation: the weight-preparation kernels of THIS step have raised the device flag if a
        weight left the f16x3 range; its value travels to pinned host memory behind the step's work and is read when that copy
        has completed - normally at the next call.  A weight that leaves the range is reported one or two steps after the
        update that produced it (the 64-step synchronising poll stays as the backstop)."""
        if ops_nn.query('xas_get_precision') != 3 or not self.opt_det.param_arena.is_cuda:
            return
        while self._range_polls and self._range_polls[0][0].query():
            _, host = self._range_polls.pop(0)
            if int(host[0]) != 0:
                self._range_polls.clear()
                raise RuntimeError('a convolution weight has left the range of the f16x3 arithmetic (|w| >= 64 or NaN) during '
                                   'the last steps: run with XAS_PRECISION=2 (bf16x6)')
        if len(self._range_polls) >= 8:               # (the host runs many steps ahead: do not pile up events)
            return
        dev = self.opt_det.param_arena.device
        word = torch.empty(1, device=dev, dtype=torch.int32)
        call('xas_f16_weight_overflow_peek', ptr(word))
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._range_polls.append((ev, host))

    def _check_loaded_weights(self):
        """Before the FIRST forward: the weights as constructed / loaded from a checkpoint must fit the f16x3 weight format -
        one scan of the detector's parameter arena (a checkpoint that does not fit must not be trained on, not even one step)."""
        if ops_nn.query('xas_get_precision') != 3 or not self.opt_det.param_arena.is_cuda:
            return
        m = float(self.opt_det.param_arena.abs().max())
        if not m < 64.0:
            raise RuntimeError('the largest detector / physique weight is %g: outside the range of the f16x3 arithmetic '
                               '(|w| < 64): run with XAS_PRECISION=2 (bf16x6)' % m)

    def __call__(self, x):
        out = {}
        loss_disc, loss_kp, total = None, {}, None
        if not self._weights_checked:
            self._check_loaded_weights()
            self._weights_checked = True
        ops_nn.reset_grad_amax()                    # (every stream of the previous step has joined this one)
        do_disc = self.opt_disc is not None and self.cur_step % self.disc_every == 0
        do_gen = self.cur_step % self.gen_every == 0
        shared = None
        if self.dedupe and do_disc and do_gen:
            ops_nn.bn_log['on'], ops_nn.bn_log['calls'] = True, []
            shared = self.model.camera_passes(x, pseudo=False)          # real-image passes, with autograd graph
            ops_nn.bn_log['on'] = False
            det_bufs = {b.data_ptr() for n, b in self.model.regressor.named_buffers() if n.endswith('running_mean')}
            ops_nn.replay_bn_updates(det_bufs)                           # the pass the discriminator step would do
            for m in self.model.regressor.modules():
                c = getattr(m, '_counters', None)
                if c is not None:
                    for _ in range(len(self.model.cam_id_list)):
                        c.bump()
        def disc_update(preds):
            ld, info = self.disc(x, self.model.regressor, preds)
            out.update(info)
            ld = ld.mean()
            if self.red_disc:
                self.red_disc.arm()
            ld.backward()
            ops_nn.join_side_stream()
            if self.red_disc:
                self.red_disc.finish()
            if self.grad_probe is not None:
                self.grad_probe('disc', self.opt_disc.grad_arena)
            self.opt_disc.step()
            self.opt_disc.zero_grad()
            return ld, info

        aux = None
        # r04: the discriminator step's detector pass as a no-grad prefix of the generator step's pass (modules/model.py:
        # joint_detector_pass) - one grouped pass of 3 * cameras groups; all of the reference's detector calls, its order
        dets = None
        joint = shared is None and do_disc and do_gen and self.model.joint_pass_possible(x)
        if joint:
            joint_preds, dets = self.model.joint_detector_pass(x)
        if do_disc:
            if shared is not None:
                loss_disc, _ = disc_update({k: v['kps'] for k, v in shared[0].items()})
            elif disc_beside_gen() and do_gen and next(self.model.regressor.parameters()).is_cuda:
                # (without the joint pass: the detector pass stays on the main chain)
                preds = joint_preds if joint else self.disc.detector_pass(x, self.model.regressor)
                main, aux = torch.cuda.current_stream(), _aux_stream()
                aux.wait_stream(main)
                with torch.cuda.stream(aux):
                    loss_disc, info = disc_update(preds)
                for t in preds.values():
                    t.record_stream(aux)
                for t in list(info.values()) + [loss_disc]:
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(main)
            else:
                loss_disc, _ = disc_update(joint_preds if joint else None)
        if do_gen:
            if shared is not None:
                self.model.pseudo_passes(x, *shared)
                loss_kp, info = self.model.finish(x, self.disc.smpl_discriminator, *shared)
            elif aux is not None and not ADV_ON_AUX:
                # the update itself stays on the second stream (beside the geometry / renderer / physique net of this pass); the
                # adversarial term - the only user of the UPDATED discriminator - runs on the main stream, after a wait placed
                # right in front of it
                loss_kp, info = self.model.finish(x, self.disc.smpl_discriminator, *self.model.camera_passes(x, dets=dets),
                                                  wait_for=aux)
            elif aux is not None:
                # the adversarial term (the only user of the UPDATED discriminator) stays on the second stream; it starts as
                # soon as the world joints exist, beside the physique net
                early = {}

                def after_geometry(per_cam):
                    if 'smpl_gen_loss' in self.model.loss_config:
                        early['v'] = self.model.adversarial_on(aux, x, self.disc.smpl_discriminator,
                                                               {k: v['world'] for k, v in per_cam.items()})
                        if 'adv_fwd' in _DEBUG_SYNC:
                            torch.cuda.synchronize()
                        if 'adv_wait' in _DEBUG_SYNC:
                            torch.cuda.current_stream().wait_stream(aux)
                cams = self.model.camera_passes(x, after_geometry=after_geometry, dets=dets)   # beside the discriminator update
                loss_kp, info = self.model.finish(x, self.disc.smpl_discriminator, *cams, aux=aux, gen_val_early=early.get('v'))
            elif dets is not None:
                loss_kp, info = self.model.finish(x, self.disc.smpl_discriminator, *self.model.camera_passes(x, dets=dets))
            else:
                loss_kp, info = self.model(x, self.disc.smpl_discriminator)
            out.update(info)
            total = sum(v.mean() for v in loss_kp.values())
            if self.red_det:
                self.red_det.arm()
            if 'pre_bwd' in _DEBUG_SYNC:
                torch.cuda.synchronize()
            total.backward()
            ops_nn.join_side_stream(reset_chains=True)
            if self.red_det:
                self.red_det.finish()
            if self.grad_probe is not None:
                self.grad_probe('det', self.opt_det.grad_arena)
            self.opt_det.step()
            self.opt_det.zero_grad()
        if aux is not None:
            torch.cuda.current_stream().wait_stream(aux)     # (already joined when the adversarial term is part of the losses)
        self.cur_step += 1
        self._poll_weight_range()
        if self.cur_step % 64 == 0:
            self._check_weight_range()               # then every 64 steps (weights drift slowly; one device synchronisation)
        return loss_disc, loss_kp, total, out
