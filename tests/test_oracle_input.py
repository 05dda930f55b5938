"""Input pipeline, CPU side: the oracle restatement and the product's host-side geometry module against goldens written
by the reference's own numpy-only loader helpers (tests/golden/make_golden.py g_input), plus internal consistency of
the restated OpenCV / scikit-fmm pieces (which are unpinned: no cv2 / skfmm in this image)."""
import numpy as np
import pytest

from conftest import golden


def test_numpy_helpers_vs_reference_golden():
    from oracle import input_pipeline as O
    from human_utils.common.imglib import affine as A
    g = golden('input_affine')
    pairs = [[1, 4], [2, 5], [3, 6], [14, 11], [15, 12], [16, 13]]
    for mod in (O, A):
        rots = [mod.norm_rot_angle(r) for r in (-540.0, -180.0, 179.5, 180.0, 181.0, 725.0)]
        assert np.array_equal(np.array(rots), g['rots'])
        r2 = np.stack([mod.rotate_2d(np.array([3.0, -2.0], dtype=np.float32), a) for a in (0.0, 0.3, -1.2, np.pi)])
        assert np.array_equal(r2, g['rot2d'])
        assert np.array_equal(mod.trans_point2d(np.array([123.0, 456.0]), g['trans']), g['pt'])
        assert np.allclose(mod.trans_points_3d(g['joints'], g['trans'], 256.0 / 2000.0), g['joints_t'], rtol=0, atol=1e-12)
        fj, fv = mod.fliplr_joints(g['joints'], g['vis'], 1000, pairs)
        assert np.array_equal(fj, g['flip_joints']) and np.array_equal(fv, g['flip_vis'])
    assert np.array_equal(O.convert_cvimg_to_tensor(g['img']), g['tensor'])
    import inputs as gi
    mask = gi.blob_mask(3, 64, seed=112)
    assert np.array_equal(np.stack([O.compute_centroid(np.bool_(m)) for m in mask]), g['centroids'])


def test_affine_from_box_properties():
    """gen_affine_trans_from_box (three-point solve): the box centre maps to the patch centre, the box edges to the patch
    edges, inv=True is the inverse map, and oracle == product module."""
    from oracle import input_pipeline as O
    from human_utils.common.imglib import affine as A
    for (cx, cy, w, h, s, r) in ((500.0, 480.0, 900.0, 900.0, 1.0, 0.0), (312.5, 700.25, 640.0, 640.0, 1.17, 23.0), (900.0, 100.0, 500.0, 500.0, 0.8, -161.0)):
        t = O.gen_affine_trans_from_box(cx, cy, w, h, 256, 256, s, r)
        assert np.allclose(t, A.gen_affine_trans_from_box_cv(cx, cy, w, h, 256, 256, s, r, False), atol=1e-12)
        assert np.allclose(O.trans_point2d((cx, cy), t), (128.0, 128.0), atol=1e-4)
        ti = O.gen_affine_trans_from_box(cx, cy, w, h, 256, 256, s, r, inv=True)
        full, fulli = np.vstack([t, [0, 0, 1]]), np.vstack([ti, [0, 0, 1]])
        assert np.allclose(full @ fulli, np.eye(3), atol=1e-5)
        assert abs(np.sqrt(abs(np.linalg.det(t[:, :2]))) - 256.0 / (w * s)) < 1e-5         # isotropic scale patch / box
        assert np.allclose(O.invert_affine(t).reshape(2, 3), ti, atol=1e-6)
        assert np.array_equal(O.invert_affine(t), A.invert_for_warp(t))


def test_warp_affine_u8_known_answers():
    from oracle import input_pipeline as O
    rng = np.random.Generator(np.random.PCG64(5))
    img = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    ident = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(O.warp_affine_u8(img, ident, 40)[:, :40], img[:40, :40])           # identity
    shift = np.array([[1.0, 0, 3.0], [0, 1.0, -2.0]])                                       # integer shift, zero border
    w = O.warp_affine_u8(img, shift, 40)
    assert np.array_equal(w[0:38, 3:40], img[2:40, 0:37]) and not w[38:, :].any() and not w[:, :3].any()
    half = np.array([[1.0, 0, 0.5], [0, 1.0, 0]])                                           # half-pixel: average of 2 neighbours
    w = O.warp_affine_u8(img[..., 0], half, 40)[..., 0]
    exp = (img[:40, 0:39, 0].astype(np.int64) + img[:40, 1:40, 0] + 1) >> 1
    assert np.array_equal(w[:, 1:40], exp.astype(np.uint8))
    const = np.full((30, 30), 200, dtype=np.uint8)                                          # weights sum to 1 exactly
    t = O.gen_affine_trans_from_box(15, 15, 12, 12, 16, 16, 1.0, 30.0)
    assert (O.warp_affine_u8(const, t, 16) == 200).all()


def test_fast_marching_known_answers():
    from oracle import input_pipeline as O
    dom = np.ones((21, 21), dtype=bool)
    src = np.zeros_like(dom)
    src[10, 10] = True
    d = O.fmm_distance(src, dom, order=1)
    assert d[10, 10] == 0 and d[10, 15] == 5.0 and d[3, 10] == 7.0                          # along the axes: exact
    assert abs(d[13, 14] - 5.0) < 0.7                                                       # first-order: overestimates diagonals
    d2 = O.fmm_distance(src, dom)                                                           # order 2 (scikit-fmm's default)
    assert d2[10, 10] == 0 and abs(d2[10, 15] - 5.0) < 1e-12 and abs(d2[3, 10] - 7.0) < 1e-12
    yy, xx = np.mgrid[0:21, 0:21]
    eu = np.hypot(yy - 10, xx - 10)
    assert np.abs(d2 - eu).max() < 0.35 < np.abs(d - eu).max()                              # second order: closer to the Euclidean distance
    assert np.abs(d2 - eu).mean() < 0.5 * np.abs(d - eu).mean()
    # second upwind neighbour not larger -> (3u - 4 v1 + v2) / 2 = 1: two pixels from the source along an axis u = 4/3 + 2/3 = 2
    line = np.ones((1, 9), dtype=bool)
    s1 = np.zeros_like(line); s1[0, 0] = True
    assert np.allclose(O.fmm_distance(s1, line)[0], np.arange(9.0), atol=1e-12)
    wall = dom.copy()
    wall[0:18, 12] = False                                                                  # obstacle: geodesic detour
    for order in (1, 2):
        dw = O.fmm_distance(src, wall, order)
        assert dw[10, 14] > d[10, 14] + 10 and dw[0, 12] == 0.0                             # masked cells report 0
    mask = np.zeros((1, 32, 32), dtype=np.float32)
    mask[0, 8:24, 10:20] = 1.0
    w, c = O.compute_geodesic_dis(mask, [2, 1, 3, 20, 0.0])
    assert c.tolist() == [[14, 15]] and w.shape == (1, 32, 32)
    assert abs(w[0, 15, 14] - (1.0 + 1.0 + 20.0)) < 1e-12                                   # centre: exp(0)+1 + 3*0+20
    assert abs(w.max() - (np.exp(2.0) + 1 + 20)) < 1e-9 or abs(w.max() - (1 + 1 + 3 + 20)) < 1e-9
    empty_center = mask.copy()
    empty_center[0, 15, 14] = 0.0
    w2, _ = O.compute_geodesic_dis(empty_center, [2, 1, 3, 20, 0.0], centers=[[14, 15]])
    assert w2.dtype == np.float16 and (w2 == 1).all()
