"""Pins the CPU oracle (oracle/) against golden vectors produced by the imported
reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

import inputs as gi
from conftest import golden
from oracle import geometry as geo
from oracle import head as ohead
from oracle import losses as L
from oracle import nets as onets
from oracle import smpl as osmpl

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, atol, rtol=0.0):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), atol=atol, rtol=rtol)


def test_head_small_forward_backward():
    g = golden('head_small')
    lg = T(g['logits']).requires_grad_(True)
    kps, dmap, idx = ohead.softargmax_multi(lg, 2, 3, 15)
    assert np.array_equal(idx.numpy(), g['z_idx'])              # int64, bit exact
    assert idx.dtype == torch.int64
    assert np.array_equal(np.sort(idx.numpy(), -1), np.sort(g['planted'], -1))
    close(kps, g['kps'], 1e-5)
    close(dmap, g['depth_prob_map'], 1e-6)
    (kps * T(g['grad_out'])).sum().backward()
    close(lg.grad, g['grad_logits'], 1e-7, 1e-4)
    k1, d1 = ohead.softargmax_single(T(g['logits']), 2)
    close(k1, g['kps_single'], 1e-5)
    close(d1, g['depth_prob_map_single'], 1e-6)


def test_head_full_size():
    g = golden('head_full')
    lg, planted = gi.planted_logits(1, 18, 64, seed=12)
    assert np.array_equal(planted, g['planted'])
    kps, dmap, idx = ohead.softargmax_multi(T(lg), 18, 3, 15)
    assert np.array_equal(idx.numpy(), g['z_idx'])
    close(kps, g['kps'], 1e-5)
    close(dmap, g['depth_prob_map'], 1e-6)
    close(ohead.softargmax_single(T(lg), 18)[0], g['kps_single'], 1e-5)


def test_links():
    g = golden('links')
    p, c = geo.skeleton_links(gi.HM36_PARENTS, gi.LINE_SELECT, False, True)
    assert p == g['parents25'].tolist() and c == g['children25'].tolist()
    p, c = geo.skeleton_links(gi.HM36_PARENTS, gi.LINE_SELECT, False, False)
    assert p == g['parents17'].tolist() and c == g['children17'].tolist()


def test_draw_lines_max():
    p, c = geo.skeleton_links(gi.HM36_PARENTS, gi.LINE_SELECT, False, True)
    for S, step in ((64, 1), (256, 4)):
        g = golden('lines_%d' % S)
        kp = T(g['kps']).requires_grad_(True)
        m = geo.draw_lines_max(kp, S, p, c, 3.0e-3)
        close(m[:, :, ::step, ::step], g['mask'], 2e-6)
        assert abs(m.double().sum().item() - g['checksum']) < 1e-3 * max(1.0, abs(g['checksum']))
        gw = T(np.random.Generator(np.random.PCG64(6)).random((2, 1, S, S)).astype(np.float32))
        (m * gw).sum().backward()
        close(kp.grad, g['grad_kps'], 2e-3, 1e-4)
    g = golden('lines_17')
    p17, c17 = geo.skeleton_links(gi.HM36_PARENTS, gi.LINE_SELECT, False, False)
    close(geo.draw_lines_max(T(g['kps']), 64, p17, c17, 3.0e-3), g['mask'], 2e-6)


def test_patch_to_world():
    g = golden('geometry')
    cam = [T(a) for a in gi.camera_params(4, seed=31)]
    kp = T(g['kps']).requires_grad_(True)
    w = geo.patch_to_world(kp, *cam)
    # world mm are O(1e3-1e4): fp32 ulp there is ~5e-4, so the bar is relative
    close(w, g['world'], 5e-3, 2e-6)
    (w * T(g['grad_out'])).sum().backward()
    close(kp.grad, g['grad_kps'], 5e-2, 1e-5)
    close(geo.patch_to_world(T(g['kps_px']), *cam, is_norm=False), g['world_px'], 5e-3, 2e-6)
    close(geo.patch_to_world(T(g['kps']), *cam, rect_width=256, mono=True, patch=False), g['world_mono'], 1e-6)


def test_losses():
    g = golden('losses')
    m, gt, w = T(g['m']), T(g['gt']), T(g['w'])
    close(L.mask_recon(m, gt), g['recon_plain'], 1e-7)
    close(L.mask_recon(m, gt, w), g['recon_w'], 1e-5)
    r = L.mask_recon(m, gt, None, True)
    assert tuple(r.shape) == (2, 1, 32, 32)                    # the non-scalar quirk
    close(r, g['recon_clip'], 1e-7)
    close(L.mask_recon(m, gt, w, True), g['recon_w_clip'], 1e-5)
    close(L.bone_sym(T(g['kp3'])), g['bone_sym'], 1e-6)
    close(L.kp_sym(T(g['kp3'])), g['kp_sym3'], 1e-6)
    close(L.kp_sym(T(g['kp2']), False), g['kp_sym2'], 1e-6)
    close(L.supervision(T(g['kp3']), T(g['kp3']).flip(0)), g['sup'], 1e-1, 1e-6)
    close(L.disc_loss(T(g['lg3']), None), g['disc_gen3'], 1e-6)
    close(L.disc_loss(T(g['lg2']), None), g['disc_gen2'], 1e-6)
    close(L.disc_loss(T(g['lg3']), T(g['gt2'])), g['disc_d'], 1e-6)


def test_physique_net():
    g = golden('physique')
    net = gi.seeded_fill_(onets.PhysiqueNet([32, 64, 128]), seed=51)
    assert list(net.state_dict().keys()) == g['keys'].tolist()
    net.train()
    x = T(g['x']).requires_grad_(True)
    y = net(x)
    close(y, g['y'], 2e-6)
    (y * T(g['grad_out'])).sum().backward()
    close(x.grad, g['grad_x'], 1e-6, 1e-4)
    close(net.encoder[0][0].weight.grad, g['grad_enc0_w'], 1e-4, 1e-4)
    close(net.decoder[4].bias.grad, g['grad_dec4_b'], 1e-4, 1e-4)
    sd = net.state_dict()
    close(sd['encoder.0.1.running_mean'], g['run_mean_enc0'], 1e-7)
    close(sd['encoder.0.1.running_var'], g['run_var_enc0'], 1e-7)
    assert int(sd['encoder.0.1.num_batches_tracked']) == int(g['nbt'])


def test_smpl():
    g = golden('smpl')
    b = {k: T(v) for k, v in gi.smpl_buffers(seed=71).items()}
    verts, jtr = osmpl.smpl_lbs(T(g['pose']), T(g['betas']), b['v_template'], b['shapedirs'], b['posedirs'],
                                b['J_regressor'], b['weights'])
    close(verts[:, ::10], g['verts_sub'], 2e-5)
    close(jtr, g['joints'], 2e-5)
    close(osmpl.smpl_to_h36m(verts, b['h36m_regressor']), g['h36m'], 2e-5)


def test_dense_to_sparse_known_answer():
    g = golden('sparse')
    ei, ea = onets.batched_dense_to_sparse(T(g['adj']))
    assert np.array_equal(ei.numpy(), g['edge_index']) and np.array_equal(ea.numpy(), g['edge_attr'])
    # the file's own example, modules/gcn.py:112-116
    assert g['edge_index'].tolist() == [[0, 0, 1, 2, 3], [0, 1, 0, 3, 3]] and g['edge_attr'].tolist() == [3, 1, 2, 1, 2]
    p17, c17 = geo.skeleton_links(gi.HM36_PARENTS, gi.LINE_SELECT, False, False)
    a = torch.eye(18).repeat(3, 1, 1)
    a[:, p17, c17] = 1.0
    a[:, c17, p17] = 1.0
    ei, ea = onets.batched_dense_to_sparse(a)
    assert ei.dtype == torch.int64
    assert np.array_equal(ei.numpy(), g['edge_index18'])
    assert ei.shape[1] == 3 * 52
    # the dense row-normalised adjacency used by the oracle/HIP path equals mean aggregation over that edge list
    adj = onets.mean_adjacency(18, p17, c17)
    x = torch.randn(18, 5)
    agg = torch.zeros(18, 5).index_add_(0, ei[1, :52], x[ei[0, :52]])
    cnt = torch.zeros(18).index_add_(0, ei[1, :52], torch.ones(52))
    close(adj @ x, agg / cnt[:, None], 1e-6)
