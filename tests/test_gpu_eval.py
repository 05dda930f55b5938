"""Evaluation path on the MI355X kernels vs the CPU oracle (oracle/evalpath.py, pinned to the reference by
tests/golden/evalpath_*.npz) - selection, triangulation, metrics, the Eval batch, and MPJPE parity end to end."""
import numpy as np
import pytest
import torch

import inputs as gi
from conftest import golden
from test_oracle_evalpath import CASES, scene

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def dev(d):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


@pytest.mark.parametrize('tag', list(CASES))
def test_select_vs_oracle(tag):
    from oracle import evalpath as ev
    from xas_amd import ops_eval
    cams, mode, x, kps = scene(tag)
    for c in cams:
        m = 'cam_%d' % c
        s3, s2, e2, sw = ev.select_hypothesis(kps[m], x[m + '_joints'], mode)
        out = ops_eval.eval_select(kps[m].cuda(), x[m + '_joints'].cuda(), mode=mode)
        assert torch.equal(out['swapped'].cpu(), sw)                       # decisions: exact
        assert torch.equal(out['sel3d'].cpu(), s3) and torch.equal(out['sel2d'].cpu(), s2)   # selections copy values
        assert float((out['err2d'].cpu() - e2).abs().max()) < 1e-6


def test_switch_points_and_mse_api():
    import eval_utils as eu
    from oracle import evalpath as ev
    cams, mode, x, kps = scene('hm36_best')
    gt = ev.normalise_gt(x['cam_1_joints'])
    p = kps['cam_1'][:, 1]
    for pts, g in ((p, gt), (p[..., :2].contiguous(), gt[..., :2].contiguous())):
        for switch_all in (False, True):
            want, wsw = ev.switch_points(pts, g, switch_all=switch_all)
            got, gsw = eu.switch_points(pts.cuda(), g.cuda(), switch_all=switch_all)
            assert gsw.shape == wsw.shape and torch.equal(gsw.cpu(), wsw)
            assert torch.equal(got.cpu(), want)
    e = eu.per_act_mse(p[..., :2].contiguous().cuda(), gt[..., :2].contiguous().cuda())
    assert float((e.cpu() - ev.per_act_mse(p[..., :2], gt[..., :2])).abs().max()) < 1e-6


@pytest.mark.parametrize('tag', ['hm36_best', 'mpi_confident'])
def test_triangulation_vs_oracle_and_fp64(tag):
    """The DLT null vector: fp32 rows exactly as the reference forms them, solved in double.  Compared with the
    oracle (torch fp32 SVD, as the reference) and with a float64 SVD of the same system: the kernel must be CLOSER
    to the float64 answer than the fp32 oracle is."""
    from modules import util as mu
    from oracle import evalpath as ev
    cams, mode, x, kps = scene(tag)
    sel = {m: ev.select_hypothesis(kps[m], x[m + '_joints'], mode)[0] for m in kps}
    want = ev.triangulation(sel, x, cams)
    xd = dev(x)
    for c in cams:
        xd['cam_%d_img' % c] = torch.empty(1, 3, 256, 256, device='cuda')
    got = mu.triangulation({m: v.cuda() for m, v in sel.items()}, xd, cams).cpu()
    # float64 reference of the same linear system
    pts = torch.stack([ev.patch_to_image(sel['cam_%d' % c], x['cam_%d_trans_image' % c], x['cam_%d_pelvis' % c]) for c in cams], 1)
    pm = torch.stack([x['cam_%d_k_mat' % c] @ torch.cat([x['cam_%d_rot_world' % c], x['cam_%d_trans_world' % c].unsqueeze(-1)], -1) for c in cams], 1)
    P0, P1, P2 = (pm[:, :, i, :].unsqueeze(1) for i in range(3))
    u, w, cf = (pts[..., [i]].permute(0, 2, 1, 3) for i in range(3))
    A = torch.cat([cf * (u * P2 - P0), cf * (w * P2 - P1)], dim=2).double()
    X = torch.linalg.svd(A)[2][:, :, -1, :]
    truth = (X / X[..., 3:])[..., :3]
    e_hip = float((got.double() - truth).norm(dim=-1).max())
    e_ora = float((want.double() - truth).norm(dim=-1).max())
    assert e_hip < 2e-3, e_hip                                  # mm, at |X| ~ 1e3: fp32 output rounding
    assert e_hip <= e_ora + 1e-6, (e_hip, e_ora)
    assert float((got - want).norm(dim=-1).max()) < max(5e-2, 4 * e_ora)
    # image points / projection matrices feeding it
    pi = mu.convert_patch_to_image(sel['cam_0'].cuda(), xd['cam_0_trans_image'], 256, 256, 256, 2000 / 256, xd['cam_0_pelvis'])
    assert float((pi.cpu() - pts[:, 0]).abs().max()) < 2e-3


def test_triangulation_recovers_planted_world():
    from modules import util as mu
    from oracle import evalpath as ev
    cams, mode, x, kps = scene('hm36_best')
    xd = dev(x)
    for c in cams:
        xd['cam_%d_img' % c] = torch.empty(1, 3, 256, 256, device='cuda')
    exact = {'cam_%d' % c: ev.normalise_gt(x['cam_%d_joints' % c]).cuda() for c in cams}
    tri = mu.triangulation(exact, xd, cams).cpu()
    assert float((tri - x['world']).norm(dim=-1).max()) < 0.5
    out = mu.batch_triangulate(torch.rand(2, 3, 5, 3).cuda() + 0.5, torch.randn(2, 3, 3, 4).cuda())
    assert out.shape == (2, 5, 4) and torch.isfinite(out).all()


def test_pose_metrics_vs_oracle():
    import metrics as M
    from oracle import evalpath as ev
    rng = np.random.Generator(np.random.PCG64(5))
    N, K = 6, 18
    gt = rng.normal(0, 400, (N, K, 3)).astype(np.float32)
    pred = (gt + rng.normal(0, 60, (N, K, 3))).astype(np.float32)
    pred[1] = (1.3 * gt[1] @ gi.random_rotation(rng).T + np.array([100., -50., 30.])).astype(np.float32)   # similarity copy
    pred[2] = gt[2] * np.array([-1., 1., 1.], np.float32)                                                   # mirrored: needs det +1 fix
    mask = rng.random((N, K)) > 0.2
    mask[0] = True
    for al in ('none', 'scale', 'procrustes'):
        want = ev.keypoint_mpjpe(pred, gt, mask, al)
        got = M.keypoint_mpjpe(pred, gt, mask, alignment=al)
        assert got.shape == want.shape and got.dtype == np.float32
        np.testing.assert_allclose(got, want, atol=2e-2, rtol=1e-4, err_msg=al)          # mm
        np.testing.assert_allclose(M.keypoint_3d_pck(pred / 1000, gt / 1000, mask, alignment=al),
                                   ev.keypoint_3d_pck(pred / 1000, gt / 1000, mask, al), atol=0)
        assert abs(M.keypoint_3d_auc(pred / 1000, gt / 1000, mask, alignment=al) -
                   ev.keypoint_3d_auc(pred / 1000, gt / 1000, mask, al)) < 1e-3
    e = M.pose_errors(pred, gt)['err'].cpu().numpy()
    assert e[2, 1].max() < 2e-2 and e[0, 1].mean() > 50                       # similarity copy: P-MPJPE ~ 0
    assert e[2, 2].mean() > 10                                                # a mirror image is NOT a rotation
    np.testing.assert_allclose(M.compute_similarity_transform(pred[3], gt[3]), ev.similarity_transform(pred[3], gt[3]),
                               atol=2e-2, rtol=1e-4)
    with pytest.raises(ValueError):
        M.keypoint_mpjpe(pred, gt, mask, alignment='affine')
    q = M.pose_errors(T(pred).cuda(), T(gt).cuda(), in_div=1000.0)
    np.testing.assert_allclose(q['pck'].cpu().numpy(), ev.keypoint_3d_pck(pred / 1000.0, gt / 1000.0, np.ones((N, K), bool)), atol=0)


def _cfg(cams, mpi=False):
    return {'model_params': {'cam_id_list': list(cams)}, 'dataset_params': {'dataset': {'name': 'mpi_inf_3dhp' if mpi else 'h36m'}}}


@pytest.mark.parametrize('tag', list(CASES))
def test_eval_batch_vs_oracle_and_golden(tag):
    """Eval.eval_batch (device) against the oracle's eval_batch and the reference-generated golden numbers."""
    import eval as xeval
    from oracle import evalpath as ev
    cams, mode, x, kps = scene(tag)
    e = xeval.Eval(_cfg(cams, mpi=True), torch.nn.Identity(), [], '/tmp')
    xd = dev(x)
    for c in cams:
        xd['cam_%d_img' % c] = torch.empty(1, 3, 256, 256, device='cuda')
    out = {k: v.cpu().numpy() for k, v in e.eval_batch(xd, mode, kps_by_cam=dev(kps)).items()}
    ora, g = ev.eval_batch(x, kps, cams, mode), golden('evalpath_' + tag)
    for ref in (ora, g):
        for k in out:
            if k.startswith('world'):
                continue
            tol = dict(atol=1e-6) if k.startswith(('err2d', 'ambiguity')) else (dict(atol=2e-3) if k.startswith(('pck', 'auc')) else dict(atol=5e-2, rtol=1e-4))
            np.testing.assert_allclose(out[k], ref[k], err_msg=k, **tol)
        np.testing.assert_allclose(out['world_gt'], ref['world_gt'], atol=2e-2, rtol=1e-5)
        # the reference's triangulation is a float32 SVD: its own result is ~0.1 mm away from the float64 solution of the
        # same system (test_triangulation_vs_oracle_and_fp64 shows the kernel is the closer of the two)
        np.testing.assert_allclose(out['world_tri'], ref['tri'], atol=0.3, rtol=1e-5)


def test_eval_loop_tables_and_result_file(tmp_path):
    """The host side: per-action tables, the synthetic loader with a consistent scene, the result file."""
    import eval as xeval

    class PlantedDetector(torch.nn.Module):
        """Returns the ground truth + a fixed offset as 3 hypotheses, keyed by the image's first pixel."""
        def forward(self, img):
            return self.kps[int(img[0, 0, 0, 0].item())], None

    for mpi in (False, True):
        cams = [0, 1, 2]
        cfg = _cfg(cams, mpi)
        cfg['train_params'] = {'batch_size': 4}
        cfg['dataset_params']['cam_id_list'] = cams
        loader = xeval.prepare_data(cfg, 1, 0, synthetic_steps=2, device=torch.device('cuda'))
        det = PlantedDetector()
        batches = list(loader)
        det.kps = {}
        for bi, x in enumerate(batches):
            for ci, c in enumerate(cams):
                m = 'cam_%d' % c
                g = x[m + '_joints'].clone()
                g[..., :2] = g[..., :2] / 255 * 2 - 1
                g[..., 2] = g[..., 2] / 255
                tagv = bi * 10 + ci
                x[m + '_img'][0, 0, 0, 0] = tagv
                det.kps[tagv] = torch.stack([g + 0.004, g + 0.05, g - 0.03], 1)
        e = xeval.Eval(cfg, det, batches, str(tmp_path))
        rec = e.eval(None, *xeval.init_tables(e.cal_per_act), mode='best')
        if mpi:
            tri = float(np.mean(rec[4]['mpjpe']) / rec[5]['mpjpe'])
        else:
            tri = float(sum(rec[4]['mpjpe'].values()) / sum(rec[5]['mpjpe'].values()))
        lines = e.record(*rec)
        txt = open(tmp_path / 'eval' / 'eval_result.txt').read()
        assert txt.strip().split('\n') == lines and any(l.startswith('2D MSE') for l in lines)
        if mpi:
            assert rec[5]['pck'] == 2 and rec[3]['auc'] == 2 * len(cams)
        else:
            assert sum(rec[1].values()) == 2 * 4 * len(cams)
        assert 0 < tri < 40                                 # the 0.004 offset (~0.5 px, ~4 mm depth): a few mm after triangulation


def test_mpjpe_parity_detector_end_to_end():
    """BASELINE metric, second half: the same synthetic evaluation batch through the HIP detector (eval-mode BN)
    and through the oracle detector gives the same MPJPE.  Tolerance: 0.5 % of the value + 0.05 mm."""
    import eval as xeval
    from oracle import evalpath as ev
    from test_gpu_model import _hip_models
    from xas_amd.synthetic import synthetic_eval_batch
    cams = [0, 1]
    reg, _, oreg, _ = _hip_models('S1', cams)
    x = synthetic_eval_batch(2, cams, torch.device('cuda'), seed=3)
    xc = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in x.items()}
    with torch.no_grad():
        for m in ('cam_0', 'cam_1'):                  # two train-mode passes: non-trivial running statistics on both sides
            reg(x[m + '_img'])
            oreg(xc[m + '_img'])
    reg.eval()
    oreg.eval()
    e = xeval.Eval(_cfg(cams), reg, [], '/tmp')
    out = {k: v.cpu().numpy() for k, v in e.eval_batch(x, 'best').items()}
    with torch.no_grad():
        okps = {m: oreg(xc[m + '_img'])[0] for m in ('cam_0', 'cam_1')}
    ora = ev.eval_batch(xc, okps, cams, 'best')
    for k in ('mpjpe_tri', 'n-mpjpe_tri', 'p-mpjpe_tri', 'mpjpe_view_cam_0', 'mpjpe_view_cam_1', 'p-mpjpe_view_cam_1'):
        np.testing.assert_allclose(out[k], ora[k], rtol=5e-3, atol=5e-2, err_msg=k)
    np.testing.assert_allclose(out['err2d_cam_0'], ora['err2d_cam_0'], rtol=5e-3, atol=1e-5)
