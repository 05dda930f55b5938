"""Patch geometry of the loader (reference: human_utils/common/imglib/affine.py:5-134), numpy on the host: these are a
handful of scalar operations per sample; the pixel work they parameterise runs on the GPU (human_utils/dataloader/gpu_patch.py).
`gen_affine_trans_from_box_cv` solves the three-point system cv2.getAffineTransform solves (no OpenCV dependency)."""
import numpy as np


def norm_rot_angle(rot):
    norm_rot = rot
    while norm_rot > 180:
        norm_rot -= 360
    while norm_rot <= -180:
        norm_rot += 360
    return norm_rot


def rotate_2d(pt_2d, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return np.array([pt_2d[0] * cs - pt_2d[1] * sn, pt_2d[0] * sn + pt_2d[1] * cs], dtype=np.float32)


def trans_point2d(pt_2d, trans):
    return np.dot(trans, np.array([pt_2d[0], pt_2d[1], 1.]).T)[0:2]


def trans_points_3d(_joints, trans, depth_scale):
    joints = _joints.copy()
    t = np.asarray(trans, dtype=np.float64)
    joints[:, 0:2] = joints[:, 0:2].astype(np.float64) @ t[:, 0:2].T + t[:, 2]       # double, as np.dot(trans, [x, y, 1])
    joints[:, 2] = joints[:, 2] * depth_scale
    return joints


def fliplr_joints(_joints, _joints_vis, width, matched_parts):
    joints, joints_vis = _joints.copy(), _joints_vis.copy()
    joints[:, 0] = width - joints[:, 0] - 1
    for a, b in matched_parts:
        joints[a, :], joints[b, :] = joints[b, :], joints[a, :].copy()
        joints_vis[a, :], joints_vis[b, :] = joints_vis[b, :], joints_vis[a, :].copy()
    return joints, joints_vis


def _affine_from_points(src, dst):
    a = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        a[i, 0:3] = [src[i][0], src[i][1], 1.0]
        a[i + 3, 3:6] = [src[i][0], src[i][1], 1.0]
        b[i], b[i + 3] = dst[i][0], dst[i][1]
    return np.linalg.solve(a, b).reshape(2, 3)


def gen_affine_trans_from_box_cv(c_x, c_y, src_width, src_height, dst_width, dst_height, scale, rot, inv):
    """Box (centre, size, rotation, scale) -> 2x3 map image -> patch (inv: patch -> image); affine.py:56-97."""
    src_w, src_h = src_width * scale, src_height * scale
    src_center = np.array([c_x, c_y], dtype=np.float32)
    rot_rad = np.pi * rot / 180
    src_downdir = rotate_2d(np.array([0, src_h * 0.5], dtype=np.float32), rot_rad)
    src_rightdir = rotate_2d(np.array([src_w * 0.5, 0], dtype=np.float32), rot_rad)
    dst_center = np.array([dst_width * 0.5, dst_height * 0.5], dtype=np.float32)
    src = np.stack([src_center, src_center + src_downdir, src_center + src_rightdir]).astype(np.float32)
    dst = np.stack([dst_center, dst_center + np.array([0, dst_height * 0.5], dtype=np.float32),
                    dst_center + np.array([dst_width * 0.5, 0], dtype=np.float32)]).astype(np.float32)
    return _affine_from_points(dst, src) if inv else _affine_from_points(src, dst)


def invert_for_warp(trans):
    """The dst -> src map cv::warpAffine derives from a forward 2x3 transform (double precision)."""
    m = np.asarray(trans, dtype=np.float64).reshape(6).copy()
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m[0], m[1], m[3], m[4] = a11, m[1] * -d, m[3] * -d, a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def trans_coords_from_patch_to_org_2d(coords_in_patch, c_x, c_y, bb_width, bb_height, rot, patch_width, patch_height):
    coords_in_org = coords_in_patch.copy()
    trans = gen_affine_trans_from_box_cv(c_x, c_y, bb_width, bb_height, patch_width, patch_height, 1.0, rot, True)
    for p in range(coords_in_patch.shape[0]):
        coords_in_org[p, 0:2] = trans_point2d(coords_in_patch[p, 0:2], trans)
    return coords_in_org


def trans_coords_from_patch_to_org_3d(coords_in_patch, c_x, c_y, bb_width, bb_height, rot, patch_width, patch_height,
                                      depth_scale):
    coords_in_org = trans_coords_from_patch_to_org_2d(coords_in_patch, c_x, c_y, bb_width, bb_height, rot, patch_width,
                                                      patch_height)
    coords_in_org[:, 2] = coords_in_patch[:, 2] * depth_scale
    return coords_in_org


# names this mirror does not replace resolve, lazily, to the reference module behind it on sys.path
from xas_amd._next import fallthrough as _fallthrough  # noqa: E402

__getattr__ = _fallthrough(__name__, __file__)
