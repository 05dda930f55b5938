"""Pins oracle/evalpath.py (selection, triangulation, metrics) against golden vectors produced by the reference's
own eval_utils / metrics / modules.util functions (tests/golden/make_golden.py: g_evalpath).  CPU only."""
import numpy as np
import pytest
import torch

import inputs as gi
from conftest import golden
from oracle import evalpath as ev

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
CASES = {'hm36_best': ([0, 1, 2, 3], 'best', 3), 'mpi_confident': ([0, 2, 4, 7, 8], 'confident', 3),
         'single': ([0, 1], 'best', 1)}


def scene(tag):
    cams, mode, hypo = CASES[tag]
    xn, kn = gi.multiview_scene(4, cams, seed=300 + len(cams) + hypo, hypo=hypo)
    return cams, mode, {k: T(v) for k, v in xn.items()}, {k: T(v) for k, v in kn.items()}


@pytest.mark.parametrize('tag', list(CASES))
def test_eval_batch_vs_reference(tag):
    cams, mode, x, kps = scene(tag)
    g = golden('evalpath_' + tag)
    out = ev.eval_batch(x, kps, cams, mode)
    assert set(out) == set(g.files)
    for k in g.files:
        if k.startswith('swapped'):
            assert np.array_equal(out[k], g[k]), k                       # bool decisions: exact
        elif k.startswith(('sel', 'err2d', 'ambiguity')):
            np.testing.assert_allclose(out[k], g[k], atol=1e-6, err_msg=k)
        elif k.startswith(('pck', 'auc')):
            np.testing.assert_allclose(out[k], g[k], atol=1e-4, err_msg=k)
        else:                                                            # world coordinates / errors in mm
            np.testing.assert_allclose(out[k], g[k], atol=2e-2, rtol=2e-5, err_msg=k)


def test_scene_is_consistent():
    """The synthetic scene is geometrically consistent: triangulating the exact ground-truth joints of all views
    returns the world joints (sub-millimetre in fp32), and the swaps planted in the detections are found."""
    cams, mode, x, kps = scene('hm36_best')
    exact = {'cam_%d' % c: ev.normalise_gt(x['cam_%d_joints' % c]) for c in cams}
    tri = ev.triangulation(exact, x, cams)
    assert float((tri - x['world']).norm(dim=-1).max()) < 0.5
    out = ev.eval_batch(x, kps, cams, mode)
    assert out['swapped_cam_0'].any() and not out['swapped_cam_0'].all()
    assert float(out['mpjpe_tri'].mean()) < float(out['mpjpe_view_cam_0'].mean())
