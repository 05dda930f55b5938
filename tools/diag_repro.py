#!/usr/bin/env python3
"""Is one optimisation step run-to-run reproducible?  Runs the SAME step (same state: parameters, moments, gradient arenas,
running statistics restored) several times in one process and compares the gradient arenas as handed to Adam, the losses and
the running statistics BIT FOR BIT; then bisects by switch (side stream off, per-camera schedule, discriminator beside the
generator off) when they are not.

  python tools/diag_repro.py [--workload HM36_Multi_SurS2] [--batch 32] [--runs 3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'x-as-supervision_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='HM36_Multi_SurS2')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--runs', type=int, default=3)
    ap.add_argument('--precision', default='f16x3')
    ap.add_argument('--p2w', action='store_true', help='diagnosis build of the library (-DXAS_P2W_DEBUG): per-lane record of patch_to_world')
    ap.add_argument('--census', default='', help='comma list of stream configurations: count differing runs of each')
    ap.add_argument('--bisect', action='store_true', help='poison the free blocks with 1e30 and report per parameter, per switch')
    ap.add_argument('--no-poison', action='store_true')
    ap.add_argument('--only-default', action='store_true')
    ap.add_argument('--poison', type=float, default=1e30)
    ap.add_argument('--loops', type=int, default=20)
    ap.add_argument('--warm', type=int, default=0, help='steps before the snapshot')
    args = ap.parse_args()
    from xas_amd import _lib as xl, engine, ops_nn
    from xas_amd.state import snapshot, restore
    from xas_amd.synthetic import model_config, synthetic_batch
    import modules.model as mm
    xl.query('xas_set_precision', xl.PREC_NAMES[args.precision])
    cfg = model_config(args.workload)
    cams = cfg['model_params']['cam_id_list']
    dev = torch.device('cuda')
    x = synthetic_batch(args.batch, cams, dev, seed=100)
    torch.manual_seed(1234)
    model, disc, od, odisc = engine.prepare_model(cfg)
    model.to(dev).train(), disc.to(dev).train()
    disc.smpl_discriminator.header.p = 0.0
    step = engine.TrainStep(cfg, model, disc, od, odisc)
    for _ in range(args.warm):
        step(x)
    torch.cuda.synchronize()
    sn = snapshot(step)

    def run(label, **sw):
        old = {}
        for k, v in sw.items():
            if k == 'foreign':
                continue
            if k == 'side':
                old[k] = ops_nn._side['enabled']
                ops_nn._side['enabled'] = v
            elif k == 'cam_batch':
                old[k] = (mm.CAM_BATCH, mm.JOIN_PSEUDO)
                mm.CAM_BATCH = mm.JOIN_PSEUDO = v
            elif k == 'joint':
                old[k] = mm.JOINT_DISC
                mm.JOINT_DISC = v
            elif k == 'beside':
                old[k] = engine._BESIDE_ENV
                engine._BESIDE_ENV = '1' if v else '0'
            elif k == 'tune':
                old[k] = 0
                xl.query('xas_set_tuning', v)
            elif k == 'adv_aux':
                old[k] = engine.ADV_ON_AUX
                engine.ADV_ON_AUX = v
        restore(step, sn)
        if args.p2w:
            import ctypes
            lib_ = xl.load()
            lib_.xas_debug_p2w.argtypes = [ctypes.c_void_p]
            lib_.xas_debug_p2w.restype = ctypes.c_int
            _foreign['p2w'] = torch.zeros(32 * 4096 * 32, device=dev)
            torch.cuda.synchronize()
            assert lib_.xas_debug_p2w(_foreign['p2w'].data_ptr()) == 0
        if sw.get('foreign'):
            # stock PyTorch kernels (elementwise over 268 MB, 4096^3 matmuls) queued on ANOTHER stream for the whole length of the
            # step: what a communication stream's kernels would be to the step - foreign work beside the library's kernels
            fs = _foreign.setdefault('stream', torch.cuda.Stream())
            if 'a' not in _foreign:
                g_ = torch.Generator(device=dev).manual_seed(3)
                _foreign['a'] = torch.randn(64, 256, 64, 64, device=dev, generator=g_)
                _foreign['m'] = torch.randn(4096, 4096, device=dev, generator=g_)
            fs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(fs):
                acc_ = torch.zeros(2, device=dev, dtype=torch.float64)
                for _ in range(int(sw['foreign'])):
                    t_ = torch.relu(_foreign['a'] * 1.0001 + 0.5)
                    u_ = _foreign['m'] @ _foreign['m']
                    acc_[0] += t_.double().sum()                 # (the foreign kernels are victims too: their results must repeat)
                    acc_[1] += u_.double().sum()
                    del t_, u_
                _foreign['acc'] = acc_
        torch.manual_seed(4242)
        grads = {}
        step.grad_probe = lambda which, arena: grads.__setitem__(which, arena.clone())
        taps = {}
        from xas_amd import ops_head
        orig_world, orig_head = mm._to_world, ops_head.softargmax_multi

        def tap_world(kps, xx, key, mono):
            if kps.requires_grad:
                kin = kps
                kps = kps.clone()                     # consumed by the geometry alone: its gradient is xas_patch_to_world_bwd's output
                taps['_saved_' + key] = kps.detach()  # (the tensor _PatchToWorld saves: re-read after the step)
                kps.register_hook(lambda g, key=key: taps.__setitem__('g_p2w_only_' + key, g.detach().clone()))
            w = orig_world(kps, xx, key, mono)
            if kps.requires_grad:
                taps['kps_clone_right_after_fwd_' + key] = kps.detach().clone()
                taps['world_reread_' + key] = w.detach().clone()
                kps = kin
                taps['world_in_' + key] = kps.detach().clone()
                taps['world_out_' + key] = w.detach().clone()
                w.register_hook(lambda g, key=key: taps.__setitem__('g_world_' + key, g.detach().clone()))
                kps.register_hook(lambda g, key=key: taps.__setitem__('g_kps_' + key, g.detach().clone()))
            return w

        def tap_head(logits, *a, **k):
            r = orig_head(logits, *a, **k)
            taps['head_kps'] = r[0].detach().clone()
            taps['head_idx'] = r[2].detach().clone()
            taps['logits_sample'] = logits.detach()[::16, ::37, ::8, ::8].clone()
            if r[0].requires_grad:
                r[0].register_hook(lambda g: taps.__setitem__('g_head_kps', g.detach().clone()))
            return r
        mm._to_world, ops_head.softargmax_multi = tap_world, tap_head
        try:
            ld, lk, tot, out = step(x)
        finally:
            mm._to_world, ops_head.softargmax_multi = orig_world, orig_head
        step.grad_probe = None
        torch.cuda.synchronize()
        res = dict(grads)
        if args.p2w:
            res['p2w'] = _foreign['p2w'].view(32, 4096, 32).clone()
        if sw.get('foreign'):
            res['foreign'] = _foreign['acc'].clone()
        for k_ in [k_ for k_ in taps if k_.startswith('_saved_')]:
            taps['saved_kps_after_step' + k_[6:]] = taps.pop(k_).clone()
        res['taps'] = taps
        res['loss'] = torch.stack([ld.detach().float().reshape(())] + [v.detach().float().mean() for v in lk.values()])
        res['bufs'] = torch.cat([b.detach().float().reshape(-1) for m in (model, disc) for b in m.buffers()])
        res['p_det'] = od.param_arena.clone()
        res['losses'] = {k: float(v.detach().mean()) for k, v in lk.items()}
        for k, v in old.items():
            if k == 'side':
                ops_nn._side['enabled'] = v
            elif k == 'cam_batch':
                mm.CAM_BATCH, mm.JOIN_PSEUDO = v
            elif k == 'joint':
                mm.JOINT_DISC = v
            elif k == 'beside':
                engine._BESIDE_ENV = v
            elif k == 'tune':
                xl.query('xas_set_tuning', 0)
            elif k == 'adv_aux':
                engine.ADV_ON_AUX = v
        return res

    _foreign = {}

    def poison(value):
        """Fill every FREE block of the caching allocator with `value` (bit pattern of a float32): take blocks of falling
        size for as long as the allocator serves them from its cache, write, release."""
        torch.cuda.synchronize()
        held = []
        size = 1 << 33
        while size >= 512:
            while True:
                before = torch.cuda.memory_reserved()
                try:
                    t = torch.empty(size // 4, device=dev, dtype=torch.float32)
                except torch.OutOfMemoryError:
                    break
                if torch.cuda.memory_reserved() > before:      # came from the driver, not from the cache: this size is used up
                    del t
                    break
                t.fill_(value)
                held.append(t)
                if len(held) > 200000:
                    break
            size >>= 1
        n = sum(t.numel() for t in held) * 4
        del held
        torch.cuda.synchronize()
        return n

    def cmp(a, b, what):
        line = []
        for k in ('det', 'disc', 'loss', 'bufs', 'p_det'):
            if k not in a:
                continue
            ne = int((a[k] != b[k]).sum())
            reln = float((a[k].double() - b[k].double()).norm() / b[k].double().norm().clamp_min(1e-300))
            extra = ''
            if k == 'det':
                flips = float(((a[k] > 0) != (b[k] > 0)).float().mean())
                extra = ' signflips %.4f' % flips
            if k == 'p_det':
                extra = ' frac|d|>1e-5 %.4f' % float(((a[k] - b[k]).abs() > 1e-5).float().mean())
            line.append('%s: %d differ, rel %.3e%s' % (k, ne, reln, extra))
        print('%-44s %s' % (what, ' | '.join(line)), flush=True)

    def beq(a_, b_):          # bit patterns (a diagnosis build that computes NaNs must still compare equal to itself)
        return torch.equal(a_.contiguous().view(torch.int32), b_.contiguous().view(torch.int32))

    base = run('default')
    print('losses', base['losses'])
    if args.census:
        CONFIGS = {'single': dict(side=False, beside=False), 'main+side': dict(side=True, beside=False),
                   'main+aux': dict(side=False, beside=True), 'main+side+aux': dict(side=True, beside=True),
                   'r04': dict(side=True, beside=True, adv_aux=True),
                   'single+foreign': dict(side=False, beside=False, foreign=60),
                   'r04+nolean': dict(side=True, beside=True, adv_aux=True, tune=262144)}
        for label in args.census.split(','):
            sw = CONFIGS[label]
            refs = [run(label, **sw) for _ in range(3)]
            # (a reference that is itself an anomaly would make every later run "differ": majority of three)
            ref = refs[0] if beq(refs[0]['det'], refs[1]['det']) or beq(refs[0]['det'], refs[2]['det']) else refs[1]
            hits, kinds, fbad = 0, {}, 0
            for it in range(args.loops):
                got = run(label, **sw)
                if 'foreign' in got and not torch.equal(got['foreign'], ref['foreign']):
                    fbad += 1
                if not (beq(got['det'], ref['det']) and beq(got['loss'], ref['loss'])):
                    hits += 1
                    if args.p2w:
                        for ck in ('cam_0', 'cam_1', 'cam_2', 'cam_3'):
                            ta, tb = got['taps'].get('world_out_' + ck), ref['taps'].get('world_out_' + ck)
                            if ta is None or torch.equal(ta, tb):
                                continue
                            slot = 4 + int(ck[-1])
                            wrong = (ta != tb).any(-1).reshape(-1).nonzero().flatten()
                            rec, recr = got['p2w'][slot], ref['p2w'][slot]
                            kin = ref['taps']['world_in_' + ck].reshape(-1, 3)
                            n_pts = kin.shape[0]
                            in_ok = (rec[:n_pts, 0:3] == kin).all(-1)
                            out_is_got = (rec[:n_pts, 3:6] == ta.reshape(-1, 3)).all(-1)
                            out_is_ref = (rec[:n_pts, 3:6] == tb.reshape(-1, 3)).all(-1)
                            print('   iteration %d %s (launch slot %d): %d wrong points %d..%d' % (it, ck, slot, wrong.numel(), int(wrong.min()), int(wrong.max())))
                            print('      at the wrong points: loaded inputs equal the true joints: %d of %d; recorded outputs equal the (wrong) memory: %d, equal the reference: %d'
                                  % (int(in_ok[wrong].sum()), wrong.numel(), int(out_is_got[wrong].sum()), int(out_is_ref[wrong].sum())))
                            print('      elsewhere: inputs true %d of %d, outputs = memory %d' % (int(in_ok.sum()) - int(in_ok[wrong].sum()), n_pts - wrong.numel(),
                                                                                                  int(out_is_got.sum()) - int(out_is_got[wrong].sum())))
                            print('      XCD of the wrong points %s; XCDs of the launch %s; pz recorded vs reference run at wrong points equal: %s' % (
                                sorted(set(rec[wrong, 6].int().tolist())), sorted(set(rec[:n_pts, 6].int().tolist())),
                                bool((rec[wrong, 7] == recr[wrong, 7]).all())))
                            names_ = ['P0', 'P1', 'P2', 'Q0', 'Q1', 'Q2', 'fx', 'fy', 'cx', 'cy'] + ['r%d' % e for e in range(9)] + ['tw0', 'tw1', 'tw2']
                            bad_fields = [names_[f] for f in range(22) if not torch.equal(rec[wrong, 8 + f], recr[wrong, 8 + f])]
                            print('      camera fields that differ from the reference run at the wrong points: %s; sample index equal: %s; HW_ID of the wrong points %s'
                                  % (bad_fields, bool(torch.equal(rec[wrong, 31], recr[wrong, 31])),
                                     sorted(set(hex(v) for v in rec[wrong, 30].view(torch.int32).tolist()))))
                            for f in bad_fields[:6]:
                                fi = 8 + names_.index(f)
                                print('         %s: got %s reference %s' % (f, rec[wrong[:4], fi].tolist(), recr[wrong[:4], fi].tolist()))
                            w0 = int(wrong[0])
                            print('      first wrong point %d: loaded %s true %s | computed %s memory %s reference %s' % (
                                w0, rec[w0, 0:3].tolist(), kin[w0].tolist(), rec[w0, 3:6].tolist(), ta.reshape(-1, 3)[w0].tolist(), tb.reshape(-1, 3)[w0].tolist()))
                    diff_taps = [k_ for k_ in ref['taps'] if k_ in got['taps'] and got['taps'][k_].shape == ref['taps'][k_].shape
                                 and not torch.equal(got['taps'][k_], ref['taps'][k_])]
                    if diff_taps and hits <= 0:
                        small = lambda d_: {k_: v_.cpu() for k_, v_ in d_.items() if v_.numel() < 200000}
                        torch.save({'got': small(got['taps']), 'ref': small(ref['taps']), 'diff': diff_taps},
                                   os.path.join(ROOT, 'gpurun_out', 'anom_%s_%d.pt' % (label.replace('+', '_'), it)))
                        print('   iteration %d: taps that differ: %s' % (it, diff_taps), flush=True)
                    d = float((got['det'].double() - ref['det'].double()).norm() / ref['det'].double().norm())
                    kinds['%.3e' % d] = kinds.get('%.3e' % d, 0) + 1
            print('== %-14s %d anomalies in %d runs  %s%s' % (label, hits, args.loops, kinds,
                                                               ('  foreign-stream checksums differing: %d' % fbad) if 'foreign' in ref else ''), flush=True)
        return
    if args.bisect:
        f = od._flat
        names = {id(p): n for n, p in list(model.named_parameters())}

        def per_param(a, b, top=25):
            rows = []
            for p, o in zip(f['params'], f['offs']):
                ga, gb = a['det'][o:o + p.numel()].double(), b['det'][o:o + p.numel()].double()
                rows.append((float((ga - gb).norm() / gb.norm().clamp_min(1e-300)), names.get(id(p), '?'), float(ga.norm()), float(gb.norm())))
            bad = [r for r in rows if r[0] > 1e-4]
            print('  parameters with gradient rel diff > 1e-4: %d of %d' % (len(bad), len(rows)))
            for r in rows:
                if r[0] > 1e-4:
                    print('    %-60s rel %.3e  |g| %.3e vs %.3e' % (r[1], r[0], r[2], r[3]))

        KEY = ('net.head.features.9.weight', 'net.head.features.9.bias', 'net.head.features.7.weight', 'net.head.features.6.weight',
               'net.head.features.3.weight', 'net.head.features.0.weight', 'net.backbone.layer4.2.conv3.weight',
               'net.backbone.layer4.2.bn3.weight', 'net.backbone.layer3.0.conv1.weight', 'net.backbone.layer1.0.conv1.weight',
               'net.backbone.bn1.weight', 'net.backbone.conv1.weight')

        def key_params(a, b):
            for p, o in zip(f['params'], f['offs']):
                n = names.get(id(p), '?').replace('regressor.', '')
                if n in KEY:
                    ga, gb = a['det'][o:o + p.numel()].double(), b['det'][o:o + p.numel()].double()
                    print('      %-44s rel %.3e  |g| %.3e vs %.3e  max|g| %.3e vs %.3e' % (
                        n, float((ga - gb).norm() / gb.norm().clamp_min(1e-300)), float(ga.norm()), float(gb.norm()),
                        float(ga.abs().max()), float(gb.abs().max())))

        labels = (('default', {}), ('beside off', dict(beside=False)), ('side off', dict(side=False)), ('joint off', dict(joint=False)))
        if args.only_default:
            labels = labels[:1]
        for label, sw in labels:
            refs = {0: run(label, **sw), 65536: run(label, tune=65536, **sw)}
            hits = 0
            for it in range(args.loops):
                if not args.no_poison:
                    poison(args.poison)
                for tune in (0, 65536):
                    got = run(label, tune=tune, **sw)
                    ref = refs[tune]
                    if not (beq(got['det'], ref['det']) and beq(got['loss'], ref['loss'])):
                        hits += 1
                        cmp(got, ref, '%s tune %d iteration %d: DIFFERS' % (label, tune, it))
                        print('      losses', ['%.6g' % float(v) for v in got['loss']], ['%.6g' % float(v) for v in ref['loss']])
                        key_params(got, ref)
                        for tk in sorted(ref['taps']):
                            ta, tb = got['taps'].get(tk), ref['taps'][tk]
                            if ta is None or ta.shape != tb.shape:
                                print('      tap %-28s missing / other shape' % tk)
                            elif not torch.equal(ta, tb):
                                d = (ta.double() - tb.double())
                                print('      tap %-28s differs: %d of %d elements, rel %.3e, per last-dim column: %s' % (
                                    tk, int((ta != tb).sum()), ta.numel(), float(d.norm() / tb.double().norm().clamp_min(1e-300)),
                                    [int(v) for v in (ta != tb).reshape(-1, ta.shape[-1]).sum(0)][:6]))
                                if ta.dim() == 4 and ta.shape[-1] == 3:
                                    idx = (ta != tb).any(-1).nonzero()[:12]
                                    for b_, h_, k_ in idx.tolist():
                                        print('          [b %d, h %d, k %d]: got %s  ref %s' % (b_, h_, k_, ['%.4e' % v for v in ta[b_, h_, k_].tolist()],
                                                                                            ['%.4e' % v for v in tb[b_, h_, k_].tolist()]))
            print('== %s: %d anomalies in %d runs' % (label, hits, 2 * args.loops), flush=True)
        return
    for i in range(args.runs - 1):
        cmp(run('default'), base, 'default run %d vs run 0' % (i + 1))
    for label, sw in (('side stream off', dict(side=False)), ('beside off', dict(beside=False)),
                      ('joint prefix off', dict(joint=False)), ('per camera', dict(cam_batch=False)),
                      ('bn 128 slabs (tune 65536)', dict(tune=65536))):
        a = run(label, **sw)
        b = run(label, **sw)
        cmp(b, a, label + ': run 1 vs run 0')
        cmp(a, base, label + ' vs default')
    for val in (float('nan'), 1e30, 0.0, -3.0):
        nb = poison(val)
        cmp(run('default'), base, 'free blocks poisoned with %r (%.1f GB)' % (val, nb / 2**30))
    for val in (float('nan'), 1e30):
        nb = poison(val)
        cmp(run('per camera', cam_batch=False), run('per camera', cam_batch=False), 'per camera, poisoned %r vs not' % val)
    # how small are the gradients Adam divides by?  |g| distribution of the detector arena
    g = base['det'].abs()
    for t in (1e-12, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6):
        print('frac |g| < %.0e: %.4f' % (t, float((g < t).float().mean())))


if __name__ == '__main__':
    main()
