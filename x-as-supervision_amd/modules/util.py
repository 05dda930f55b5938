"""Geometry helpers of the training path on the MI355X kernels (reference: modules/util.py).

Built here: `draw_lines` (+ the fused `draw_lines_max` the model actually consumes),
`convert_patch_to_world` (all hypotheses in one launch, closed-form 2x2 / 3x3 inverses),
`make_coordinate_grid`, `smpl_to_h36m`, and for the evaluation path `convert_patch_to_image`,
`triangulation` / `batch_triangulate` (device DLT, no batched SVD).  The reference's dead code
(rule_transformation, my_truncated_normal, project_smpl_to_patch_kps has no caller) is not rebuilt.
"""
import torch

from xas_amd import ops_eval, ops_head


def make_coordinate_grid(spatial_size, type):
    """[-1,1] x [-1,1] mesh, grid[i, j] = (2j/(w-1)-1, 2i/(h-1)-1)   (util.py:3-19)."""
    h, w = spatial_size
    xs = 2 * (torch.arange(w).type(type) / (w - 1)) - 1
    ys = 2 * (torch.arange(h).type(type) / (h - 1)) - 1
    return torch.stack([xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w)], dim=2)


def draw_lines_max(keypoints, image_size, parent_ids, child_ids, body_width):
    """max over the line heat-maps of `draw_lines`, [B,1,S,S]: the only way the model uses them
    (model.py:91-96).  One fused kernel; nothing but the mask is written."""
    return ops_head.draw_lines_max(keypoints, image_size, parent_ids, child_ids, body_width)


def draw_lines(keypoints, image_size, parent_ids, child_ids, body_width):
    """Per-line heat-maps [B, L, S, S] (util.py:21-59).  The training step never materialises them
    (see draw_lines_max); this entry renders one line per launch for callers that want the stack."""
    maps = []
    fine = (11, 12, 14, 15) if len(parent_ids) >= 21 else ()
    for l, (p, c) in enumerate(zip(parent_ids, child_ids)):
        width = body_width / 2.0 if l in fine else body_width      # exponent x2 == width / 2
        maps.append(ops_head.draw_lines_max(keypoints, image_size, [p], [c], width))
    return torch.cat(maps, dim=1)


def convert_patch_to_world(keypoints, params, mode, is_norm=True, RECT_WIDTH=2000, mono=False, patch=True):
    """Patch coordinates -> world mm for camera `mode` of the batch dict (util.py:128-152).
    keypoints: [B,K,3] or, for all hypotheses at once, [B,Hy,K,3]."""
    size = params['{}_img'.format(mode)].shape[-1]
    return ops_head.patch_to_world(
        keypoints, params['{}_trans_image'.format(mode)], params['{}_k_mat'.format(mode)],
        params['{}_pelvis'.format(mode)], params['{}_rot_world'.format(mode)], params['{}_trans_world'.format(mode)],
        image_size=size, rect_width=RECT_WIDTH, is_norm=is_norm, mono=mono, patch=patch)


def convert_patch_to_image(kps, trans, image_depth, image_height, image_width, depth_scale, pelvis, is_norm=True):
    """Patch -> image (u px, v px, depth mm), util.py:61-83.  The kernel takes the square patch size and the
    metric box width; `depth_scale` is RECT_WIDTH / image size at every call site (util.py:180-182)."""
    if not (image_depth == image_height == image_width):
        raise RuntimeError('convert_patch_to_image: the patch must be a cube (D == H == W)')
    return ops_eval.patch_to_image(kps, trans, pelvis, image_size=image_width, rect_width=depth_scale * image_width,
                                   is_norm=is_norm)


def batch_triangulate(keypoints_, Pall):
    """keypoints_ [B,V,K,3] = (u, v, weight), Pall [B,V,3,4] -> [B,K,4] (util.py:198-230)."""
    return ops_eval.triangulate_dlt(keypoints_, Pall)


def triangulation(keypoints, params, cam_id_list, is_norm=True, RECT_WIDTH=2000):
    """{cam_key: [B,K,3]} patch predictions of all cameras -> world joints [B,K,3] (util.py:171-196)."""
    pts, pm = [], []
    for cam_id in cam_id_list:
        mode = 'cam_{}'.format(cam_id)
        size = params['{}_img'.format(mode)].shape[-1]
        pts.append(ops_eval.patch_to_image(keypoints[mode], params['{}_trans_image'.format(mode)],
                                           params['{}_pelvis'.format(mode)], image_size=size, rect_width=RECT_WIDTH,
                                           is_norm=is_norm))
        pm.append(ops_eval.projection_matrix(params['{}_k_mat'.format(mode)], params['{}_rot_world'.format(mode)],
                                             params['{}_trans_world'.format(mode)]))
    return batch_triangulate(torch.stack(pts, dim=1), torch.stack(pm, dim=1))[..., :3]


def smpl_to_h36m(verts, h36m_regressor):
    """17 regressed joints, L/R arms swapped, thorax appended, root centred (util.py:331-341)."""
    j = torch.einsum('bki,lk->bli', verts, h36m_regressor)
    j = j[:, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14, 15, 16, 11, 12, 13]]
    j = torch.cat([j, j[:, [11, 14]].mean(dim=1, keepdim=True)], dim=1)
    return j - j[:, 0:1]


def random_rotation_3D(keypoints):
    """Random rotation about z in [-pi/4, pi/4] per sample (util.py:389-407; only with use_aug).  The angles are drawn
    from the CPU generator exactly as the reference does (torch.rand(B, 1)): same seed, same augmentation."""
    B = keypoints.shape[0]
    ang = ((torch.rand(B, 1) - 0.5) * 0.5 * torch.pi).squeeze(1).to(keypoints.device)
    c, s, z, o = torch.cos(ang), torch.sin(ang), torch.zeros_like(ang), torch.ones_like(ang)
    rot = torch.stack([c, -s, z, s, c, z, z, z, o], dim=1).view(B, 3, 3)
    return torch.bmm(keypoints, rot.to(keypoints.dtype))


# names this mirror does not replace resolve, lazily, to the reference module behind it on sys.path
from xas_amd._next import fallthrough as _fallthrough  # noqa: E402

__getattr__ = _fallthrough(__name__, __file__)
