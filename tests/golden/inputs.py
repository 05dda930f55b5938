"""Deterministic input generators shared by ``make_golden.py`` (which feeds them to the
imported reference) and by the tests (which feed them to the oracle / the HIP path).
Only numpy PCG64 and closed forms are used so every machine with this image
reproduces the same bits.  Nothing here comes from the reference."""
import numpy as np

HM36_PARENTS = [0, 0, 1, 2, 0, 4, 5, 0, 17, 8, 9, 17, 11, 12, 17, 14, 15, 7]
LINE_SELECT = list(range(17))


def planted_logits(B, K, D, seed, peaks=None, noise=0.05):
    """Logits [B, K*D, D, D] whose softmax has three well separated depth peaks per
    joint (so top-k indices are unambiguous) plus a smooth x/y bump and small noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = np.arange(D, dtype=np.float64)
    out = np.empty((B, K, D, D, D), np.float32)
    planted = np.empty((B, K, 3), np.int64)
    for b in range(B):
        for k in range(K):
            if peaks is None:
                # three depth centres at least D/5 apart, away from the borders
                base = rng.permutation(3)
                cz = np.array([D * 0.2, D * 0.5, D * 0.8]) + rng.uniform(-D * 0.04, D * 0.04, 3)
                cz = np.round(cz[base])
            else:
                cz = np.asarray(peaks, np.float64)
            amp = np.array([6.0, 5.0, 4.0])
            cx, cy = rng.uniform(D * 0.25, D * 0.75, 2)
            sz, sxy = D / 32.0 + 0.6, D / 10.0
            fz = sum(a * np.exp(-0.5 * ((g - c) / sz) ** 2) for a, c in zip(amp, cz))
            fxy = np.exp(-0.5 * (((g[None, :] - cx) / sxy) ** 2 + ((g[:, None] - cy) / sxy) ** 2))
            vol = fz[:, None, None] + 3.0 * fxy[None, :, :]
            vol = vol + noise * rng.standard_normal((D, D, D))
            out[b, k] = vol.astype(np.float32)
            planted[b, k] = cz.astype(np.int64)
    return out.reshape(B, K * D, D, D), planted


def skeleton_2d(B, seed, K=18):
    """Plausible normalised 2-D joints [B,K,2] in [-0.8, 0.8]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.uniform(-0.8, 0.8, (B, K, 2)).astype(np.float32)


def random_rotation(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def camera_params(B, seed):
    """trans_image [B,2,3], k_mat [B,3,3], pelvis [B,3], rot_world [B,3,3], trans_world [B,3]
    with the magnitudes of human_utils/dataloader/dataloader.py:166-191."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ti = np.zeros((B, 2, 3))
    km = np.zeros((B, 3, 3))
    for b in range(B):
        s = rng.uniform(0.24, 0.32)
        th = rng.uniform(-0.2, 0.2)
        ti[b, :, :2] = s * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        ti[b, :, 2] = rng.uniform(-40, 40, 2)
        km[b] = np.diag([rng.uniform(1100, 1200), rng.uniform(1100, 1200), 1.0])
        km[b, 0, 2], km[b, 1, 2] = rng.uniform(480, 540, 2)
    pelvis = np.stack([rng.uniform(-500, 500, B), rng.uniform(-500, 500, B), rng.uniform(4000, 6000, B)], 1)
    rot = np.stack([random_rotation(rng) for _ in range(B)])
    tw = rng.uniform(-3000, 3000, (B, 3))
    f32 = lambda a: a.astype(np.float32)
    return f32(ti), f32(km), f32(pelvis), f32(rot), f32(tw)


def blob_mask(B, S, seed):
    """Binary body-like masks [B,1,S,S]: union of discs along a random poly-line."""
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:S, 0:S]
    m = np.zeros((B, 1, S, S), np.float32)
    for b in range(B):
        pts = rng.uniform(0.25 * S, 0.75 * S, (6, 2))
        for i in range(5):
            for t in np.linspace(0, 1, 12):
                c = pts[i] * (1 - t) + pts[i + 1] * t
                m[b, 0][(xx - c[0]) ** 2 + (yy - c[1]) ** 2 < (0.06 * S) ** 2] = 1.0
    return m


def synthetic_batch(B, cam_ids, seed=0, S=256, K=18):
    """The batch-dict contract of human_utils/dataloader/dataloader.py:166-191,221,228
    filled with synthetic tensors.  Returns a dict of numpy arrays."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = {}
    for ci, cam in enumerate(cam_ids):
        key = 'cam_%s' % cam
        mask = blob_mask(B, S, seed * 100 + ci)
        x[key + '_mask'] = mask
        x[key + '_img'] = (rng.random((B, 3, S, S), dtype=np.float32) * mask).astype(np.float32)
        x[key + '_geodesic_dis'] = (1.0 + 24.0 * rng.random((B, 1, S, S), dtype=np.float32)).astype(np.float32)
        j = rng.uniform(40, 216, (B, K, 3)).astype(np.float32)
        j[..., 2] = rng.uniform(-40, 40, (B, K))
        x[key + '_joints'] = j
        ti, km, pv, rw, tw = camera_params(B, seed * 1000 + 17 * ci + 1)
        x[key + '_trans_image'], x[key + '_k_mat'], x[key + '_pelvis'] = ti, km, pv
        x[key + '_rot_world'], x[key + '_trans_world'] = rw, tw
        pmask = blob_mask(B, S, seed * 100 + 50 + ci)
        x[key + '_pseudo_img'] = (rng.random((B, 3, S, S), dtype=np.float32) * pmask).astype(np.float32)
        pj = rng.uniform(-0.8, 0.8, (B, K, 3)).astype(np.float32)
        pj[..., 2] = rng.uniform(-0.4, 0.4, (B, K))
        x[key + '_pseudo_joints'] = pj
    return x


def seeded_fill_(module, seed, gain=1.0):
    """Deterministically (re)initialise every parameter / BN buffer of a torch module:
    conv / linear weights ~ N(0, 2/fan_in), biases ~ N(0, 0.05), norm gamma ~ U(0.6,1.4),
    norm beta ~ N(0, 0.1).  Uses a CPU torch.Generator (bit-reproducible for a fixed
    torch build); values are produced in state-dict key order."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if not t.dtype.is_floating_point:
                continue
            leaf = name.rsplit('.', 1)[-1]
            if leaf == 'running_mean':
                t.zero_()
            elif leaf == 'running_var':
                t.fill_(1.0)
            elif t.dim() >= 2:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * (gain * (2.0 / fan_in) ** 0.5))
            elif leaf == 'weight':
                t.copy_(0.6 + 0.8 * torch.rand(t.shape, generator=g))
            else:
                t.copy_(0.1 * torch.randn(t.shape, generator=g))
    return module


def planted_depth_bias(K, D, seed):
    """Bias [K*D] for the final 1x1 conv that plants three separated depth peaks per joint."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = np.arange(D, dtype=np.float64)
    out = np.zeros((K, D))
    for k in range(K):
        cz = np.round(np.array([D * 0.2, D * 0.5, D * 0.8]) + rng.uniform(-D * 0.04, D * 0.04, 3))
        amp = np.array([6.0, 5.0, 4.0])[rng.permutation(3)]
        out[k] = sum(a * np.exp(-0.5 * ((g - c) / (D / 32.0 + 0.6)) ** 2) for a, c in zip(amp, cz))
    return out.reshape(-1).astype(np.float32)


def smpl_buffers(seed, V=6890):
    """Synthetic stand-ins for the licensed SMPL arrays, shapes of smpl_layer.py:40-55."""
    rng = np.random.Generator(np.random.PCG64(seed))
    f32 = lambda a: a.astype(np.float32)
    w = rng.random((V, 24)) ** 8
    w /= w.sum(1, keepdims=True)
    jr = rng.random((24, V)) ** 6
    jr /= jr.sum(1, keepdims=True)
    hr = rng.random((17, V)) ** 6
    hr /= hr.sum(1, keepdims=True)
    return dict(v_template=f32(rng.uniform(-1, 1, (1, V, 3))),
                shapedirs=f32(0.03 * rng.standard_normal((V, 3, 10))),
                posedirs=f32(0.01 * rng.standard_normal((V, 3, 207))),
                J_regressor=f32(jr), weights=f32(w), h36m_regressor=f32(hr))

def seeded_state_dict(keys, shapes, seed, gain=1.0):
    """The values seeded_fill_ gives a module whose state dict has exactly these keys / shapes IN THIS ORDER
    (used to load a golden's parameter set into a module whose registration order may differ)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in zip(keys, shapes):
        shape = tuple(shape)
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            continue
        if leaf == 'running_mean':
            out[name] = torch.zeros(shape)
        elif leaf == 'running_var':
            out[name] = torch.ones(shape)
        elif len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            out[name] = torch.randn(shape, generator=g) * (gain * (2.0 / fan_in) ** 0.5)
        elif leaf == 'weight':
            out[name] = 0.6 + 0.8 * torch.rand(shape, generator=g)
        else:
            out[name] = 0.1 * torch.randn(shape, generator=g)
    return out


def model_params(stage='S1', cam_ids=(0, 1, 2, 3)):
    """model_params section of config/HM36_Multi_SurS1.yaml:30-87 / SurS2 (values only)."""
    s2 = stage == 'S2'
    lc = {'recons_loss': {'use_dis_map': not s2, 'weight': 0.02 if s2 else 0.0},
          'physique_recons_loss': {'use_dis_map': not s2, 'weight': 0.02 if s2 else 0.0},
          'smpl_pseudo_img_loss': {'weight': 3.0}}
    if s2:
        lc['symmetry_loss'] = {'weight': {'bone': 0.1, 'kp': 0.1, 'kp_2d': 0.0}}
    lc['smpl_disc_loss'] = {'weight': 0.5 if s2 else 0.0, 'update_interval': 1}
    lc['smpl_gen_loss'] = {'weight': 0.5 if s2 else 0.0}
    return {
        'detector_params': {'name': 'resnet_multi', 'num_kp': 18, 'depth_dim': 64, 'num_hypo': 3, 'neighbor_size': 15},
        'smpl_disc_params': {'name': 'res_sage_gcn_decouple', 'input_dim': 128, 'hidden_dim': 128, 'output_dim': 128,
                             'num_node': 18, 'disc_sup_dim': 3, 'num_layers': 2, 'use_self_loop': True, 'use_pe': True},
        'smpl_layer_params': {'model_path': 'data/smpl_models'},
        'physique_mask_generator_params': {'layers': [32, 64, 128]},
        'parent_ids': list(HM36_PARENTS), 'child_ids': list(range(18)),
        'flip_pairs': [[1, 4], [2, 5], [3, 6], [14, 11], [15, 12], [16, 13]],
        'line_select_ids': list(LINE_SELECT), 'body_width': 3.0,
        'loss_config': lc, 'cam_id_list': list(cam_ids)}


def multiview_scene(B, cam_ids, seed, K=18, S=256, rect=2000.0, hypo=3, noise=0.01):
    """A geometrically CONSISTENT multi-camera scene for the evaluation path: world joints projected into every
    camera (pin-hole, crop affine, depth px), plus noisy multi-hypothesis 'detections' with some left/right swaps.
    Returns (x, kps): x = batch dict of numpy arrays (camera parameters + `_joints` in pixel units, `world`),
    kps = {cam_key: [B,hypo,K,3]} in the detector's normalised output range."""
    rng = np.random.Generator(np.random.PCG64(seed))
    world = rng.normal(0.0, 350.0, (B, K, 3))
    world[:, 0] = rng.normal(0.0, 50.0, (B, 3))
    x, kps = {'world': world.astype(np.float32)}, {}
    pairs = ((1, 4), (2, 5), (3, 6), (14, 11), (15, 12), (16, 13))
    for cam in cam_ids:
        key = 'cam_%s' % cam
        ti, km, pv, rw, tw = (np.zeros((B, 2, 3)), np.zeros((B, 3, 3)), np.zeros((B, 3)), np.zeros((B, 3, 3)),
                              np.zeros((B, 3)))
        joints = np.zeros((B, K, 3))
        for b in range(B):
            R = random_rotation(rng)
            t = np.array([rng.uniform(-300, 300), rng.uniform(-300, 300), rng.uniform(4500, 5500)])
            camp = world[b] @ R.T + t
            fx, fy, cx, cy = rng.uniform(1100, 1200), rng.uniform(1100, 1200), rng.uniform(480, 540), rng.uniform(480, 540)
            u, v = camp[:, 0] / camp[:, 2] * fx + cx, camp[:, 1] / camp[:, 2] * fy + cy
            s, th = rng.uniform(0.24, 0.32), rng.uniform(-0.2, 0.2)
            A = s * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
            off = np.array([S / 2, S / 2]) + rng.uniform(-6, 6, 2) - A @ np.array([u[0], v[0]])
            uv = np.stack([u, v], 1) @ A.T + off
            joints[b, :, :2] = uv
            joints[b, :, 2] = (camp[:, 2] - camp[0, 2]) / (rect / S)
            ti[b, :, :2], ti[b, :, 2] = A, off
            km[b] = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
            pv[b], rw[b], tw[b] = camp[0], R, t
        f32 = lambda a: a.astype(np.float32)
        x[key + '_joints'], x[key + '_trans_image'], x[key + '_k_mat'] = f32(joints), f32(ti), f32(km)
        x[key + '_pelvis'], x[key + '_rot_world'], x[key + '_trans_world'] = f32(pv), f32(rw), f32(tw)
        g = joints.copy()
        g[..., :2] = g[..., :2] / (S - 1) * 2 - 1
        g[..., 2] = g[..., 2] / (S - 1)
        det = np.repeat(g[:, None], hypo, 1) + rng.normal(0, noise, (B, hypo, K, 3))
        for h in range(1, hypo):
            det[:, h, :, 2] += rng.normal(0, 0.08, (B, K))              # wrong-depth hypotheses
        for b in range(B):                                              # mirrored detections: some joints, some hypotheses
            for h in range(hypo):
                for a, c in pairs:
                    if rng.random() < 0.3:
                        det[b, h, [a, c]] = det[b, h, [c, a]]
        kps[key] = f32(det)
    return x, kps
