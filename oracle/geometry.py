"""Oracle: skeleton links, patch->world geometry, line-mask renderer.
TEST INFRASTRUCTURE ONLY.

Follows modules/model.py:8-22 (cal_links), modules/util.py:61-95,128-152
(convert_patch_to_world and helpers) and modules/util.py:3-59 + the two
torch.max at modules/model.py:94,96 (draw_lines followed by max over lines).
"""
import torch

EXT_PARENTS = (7, 7, 7, 7, 0, 0, 1, 4)      # modules/model.py:19
EXT_CHILDREN = (1, 4, 11, 14, 2, 5, 14, 11)  # modules/model.py:20
FINE_LINES = (11, 12, 14, 15)               # modules/util.py:52 (arms use thin lines)


def skeleton_links(parent_ids, line_select_ids=None, use_root=False, extension=True):
    """(parents, children) index lists.  modules/model.py:8-22."""
    if use_root:
        children = list(range(len(parent_ids)))
        parents = list(parent_ids)
    else:
        children = list(range(1, len(parent_ids)))
        parents = list(parent_ids[1:])
    parents = [parents[i] for i in line_select_ids]
    children = [children[i] for i in line_select_ids]
    if extension:
        parents += list(EXT_PARENTS)
        children += list(EXT_CHILDREN)
    return parents, children


def patch_to_world(kps, trans_image, k_mat, pelvis, rot_world, trans_world,
                   image_size=256, rect_width=2000.0, is_norm=True, mono=False, patch=True):
    """kps [B,K,3] -> world [B,K,3].

    patch=True  : undo normalisation ((x+1)/2*(S-1), z*(S-1)), undo the 2x3 crop
                  affine, depth px -> mm (+ pelvis z)          util.py:61-83
    mono=False  : pin-hole back projection and R^-1 (p - T)     util.py:86-95
    mono=True   : visualisation branch: z += 128, reorder [x,z,y], negate
                                                                util.py:145-150
    """
    S = float(image_size)
    p = kps
    if patch:
        if is_norm:
            u = (p[..., 0] + 1) / 2.0 * (S - 1)
            v = (p[..., 1] + 1) / 2.0 * (S - 1)
            z = p[..., 2] * (S - 1)
        else:
            u, v, z = p[..., 0], p[..., 1], p[..., 2]
        A = trans_image[:, :, :2]
        t = trans_image[:, :, 2]
        det = A[:, 0, 0] * A[:, 1, 1] - A[:, 0, 1] * A[:, 1, 0]
        inv = torch.stack([torch.stack([A[:, 1, 1], -A[:, 0, 1]], -1),
                           torch.stack([-A[:, 1, 0], A[:, 0, 0]], -1)], 1) / det.view(-1, 1, 1)
        du = u - t[:, 0:1]
        dv = v - t[:, 1:2]
        u2 = inv[:, 0, 0:1] * du + inv[:, 0, 1:2] * dv
        v2 = inv[:, 1, 0:1] * du + inv[:, 1, 1:2] * dv
        zc = z * (1.0 / S * rect_width) + pelvis[:, 2:3]
        p = torch.stack([u2, v2, zc], dim=-1)
    if mono:
        return -torch.stack([p[..., 0], p[..., 2] + 128, p[..., 1]], dim=-1)
    fx, fy = k_mat[:, 0, 0:1], k_mat[:, 1, 1:2]
    cx, cy = k_mat[:, 0, 2:3], k_mat[:, 1, 2:3]
    zc = p[..., 2]
    cam = torch.stack([(p[..., 0] - cx) / fx * zc, (p[..., 1] - cy) / fy * zc, zc], dim=-1)
    rinv = torch.linalg.inv(rot_world.double()).to(kps.dtype)
    return torch.einsum('bij,bkj->bki', rinv, cam - trans_world.unsqueeze(1))


def line_heatmaps(kps2d, image_size, parents, children, body_width):
    """[B,N,2] in [-1,1] -> per-line heat-maps [B,L,S,S].  modules/util.py:21-59.
    Grid point (row i, col j) = (2j/(S-1)-1, 2i/(S-1)-1)  (util.py:8-17)."""
    B = kps2d.shape[0]
    S = image_size
    lin = 2 * (torch.arange(S, dtype=kps2d.dtype) / (S - 1)) - 1
    gx = lin.view(1, 1, 1, S).expand(1, 1, S, S)
    gy = lin.view(1, 1, S, 1).expand(1, 1, S, S)
    a = kps2d[:, children]            # start  [B,L,2]
    b = kps2d[:, parents]             # end
    v = b - a
    ax, ay = a[..., 0, None, None], a[..., 1, None, None]
    bx, by = b[..., 0, None, None], b[..., 1, None, None]
    vx, vy = v[..., 0, None, None], v[..., 1, None, None]
    dax, day = gx - ax, gy - ay
    t = (dax * vx + day * vy) / (1e-8 + vx * vx + vy * vy)
    d_start = dax * dax + day * day
    d_end = (gx - bx) ** 2 + (gy - by) ** 2
    d_mid = (gx - (ax + t * vx)) ** 2 + (gy - (ay + t * vy)) ** 2
    d2 = torch.where(t <= 0, d_start, torch.where(t >= 1, d_end, d_mid))
    e = -d2 / body_width
    if e.shape[1] >= 21:
        scale = torch.ones(e.shape[1], dtype=e.dtype)
        scale[list(FINE_LINES)] = 2.0
        e = e * scale.view(1, -1, 1, 1)
    return torch.exp(e)


def draw_lines_max(kps2d, image_size, parents, children, body_width):
    """max over lines -> [B,1,S,S]  (modules/model.py:91-96)."""
    return line_heatmaps(kps2d, image_size, parents, children, body_width).max(dim=1, keepdim=True)[0]


def random_rotation_3d(kp):
    """modules/util.py:389-407: rotation about z by an angle in [-pi/4, pi/4] per sample; the angles come from the CPU
    generator (torch.rand(B, 1)), one draw per call."""
    B = kp.shape[0]
    ang = ((torch.rand(B, 1) - 0.5) * 0.5 * torch.pi).squeeze()
    rot = torch.zeros(B, 3, 3)
    rot[:, 0, 0] = torch.cos(ang)
    rot[:, 0, 1] = -torch.sin(ang)
    rot[:, 1, 0] = torch.sin(ang)
    rot[:, 1, 1] = torch.cos(ang)
    rot[:, 2, 2] = 1
    return torch.bmm(kp.clone(), rot.to(kp.dtype))
