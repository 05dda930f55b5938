"""Oracle: the reference's CPU input pipeline per sample and camera.  TEST INFRASTRUCTURE ONLY.

Follows human_utils/dataloader/dataloader.py:17-91,150-191 (`generate_patch_sample_data`, `generate_item`),
human_utils/common/imglib/affine.py:14-114, human_utils/common/imglib/format.py:4-12 and
human_utils/common/utility/geodesic.py:4-54 in plain numpy.

Third-party arithmetic, absent from this image and from /root/reference (requirements.txt:7,14, unpinned versions):
  * opencv-python: `cv2.getAffineTransform`, `cv2.warpAffine(..., INTER_LINEAR)` on 8-bit images, `cv2.GaussianBlur(5x5)`
    + `cv2.threshold`.  Restated from OpenCV 4.x's published algorithm (modules/imgproc/src/imgwarp.cpp: AB_BITS = 10,
    INTER_BITS = 5, INTER_REMAP_COEF_BITS = 15, BORDER_CONSTANT 0; smooth: fixed-point [1 4 6 4 1]/16 kernel,
    BORDER_REFLECT_101).  PARITY UNPINNED: no cv2 here to generate vectors from.
  * scikit-fmm: `skfmm.distance` (fast marching).  Restated as a heap fast-marching solve of the library's default
    SECOND-order scheme (order=2: the one-sided second-order difference where the second upwind neighbour is frozen and not
    larger, first order elsewhere; tentative values re-computed whenever a neighbour freezes), r05; order=1 (the r03-r04
    restatement) is kept.  PARITY UNPINNED: the package is not in the image.
The reference's own numpy-only helpers (rotate_2d, trans_point2d, trans_points_3d, fliplr_joints, norm_rot_angle,
convert_cvimg_to_tensor, compute_centroid) ARE pinned by tests/golden/input_affine.npz (make_golden.py g_input)."""
import heapq

import numpy as np


# ------------------------------------------------------------------ affine.py (numpy-only pieces)
def norm_rot_angle(rot):
    while rot > 180:
        rot -= 360
    while rot <= -180:
        rot += 360
    return rot


def rotate_2d(pt, rot_rad):
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return np.array([pt[0] * cs - pt[1] * sn, pt[0] * sn + pt[1] * cs], dtype=np.float32)


def trans_point2d(pt, trans):
    return np.dot(trans, np.array([pt[0], pt[1], 1.0]).T)[0:2]


def trans_points_3d(joints, trans, depth_scale):
    out = joints.copy()
    for j in range(len(out)):
        out[j, 0:2] = trans_point2d(out[j, 0:2], trans)
        out[j, 2] = out[j, 2] * depth_scale
    return out


def fliplr_joints(joints, vis, width, pairs):
    joints, vis = joints.copy(), vis.copy()
    joints[:, 0] = width - joints[:, 0] - 1
    for a, b in pairs:
        joints[a, :], joints[b, :] = joints[b, :], joints[a, :].copy()
        vis[a, :], vis[b, :] = vis[b, :], vis[a, :].copy()
    return joints, vis


def get_affine_transform(src, dst):
    """cv2.getAffineTransform: the 2x3 map taking three points src[i] to dst[i] (6x6 linear system, double)."""
    a = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        a[i, 0:3] = [src[i][0], src[i][1], 1.0]
        a[i + 3, 3:6] = [src[i][0], src[i][1], 1.0]
        b[i], b[i + 3] = dst[i][0], dst[i][1]
    return np.linalg.solve(a, b).reshape(2, 3)


def gen_affine_trans_from_box(c_x, c_y, src_width, src_height, dst_width, dst_height, scale, rot, inv=False):
    """affine.py:56-97."""
    src_w, src_h = src_width * scale, src_height * scale
    center = np.array([c_x, c_y], dtype=np.float32)
    rad = np.pi * rot / 180
    down = rotate_2d(np.array([0, src_h * 0.5], dtype=np.float32), rad)
    right = rotate_2d(np.array([src_w * 0.5, 0], dtype=np.float32), rad)
    dcen = np.array([dst_width * 0.5, dst_height * 0.5], dtype=np.float32)
    src = np.stack([center, center + down, center + right]).astype(np.float32)
    dst = np.stack([dcen, dcen + np.array([0, dst_height * 0.5], dtype=np.float32),
                    dcen + np.array([dst_width * 0.5, 0], dtype=np.float32)]).astype(np.float32)
    return get_affine_transform(dst, src) if inv else get_affine_transform(src, dst)


def invert_affine(m):
    """The inversion cv::warpAffine applies to a forward 2x3 map (imgwarp.cpp), in double."""
    m = np.asarray(m, dtype=np.float64).reshape(6).copy()
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m[0], m[1], m[3], m[4] = a11, m[1] * -d, m[3] * -d, a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def warp_affine_u8(img, trans, P):
    """cv2.warpAffine(img, trans, (P, P), flags=INTER_LINEAR), 8-bit, BORDER_CONSTANT 0 (vectorised numpy, exact ints)."""
    img = np.asarray(img)
    if img.ndim == 2:
        img = img[..., None]
    H, W, C = img.shape
    m = invert_affine(trans)
    x = np.arange(P, dtype=np.float64)
    y = np.arange(P, dtype=np.float64)
    adelta = np.rint(m[0] * x * 1024).astype(np.int64)
    bdelta = np.rint(m[3] * x * 1024).astype(np.int64)
    X0 = np.rint((m[1] * y + m[2]) * 1024).astype(np.int64) + 16
    Y0 = np.rint((m[4] * y + m[5]) * 1024).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    out = np.zeros((P, P, C), dtype=np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64) * ok[..., None]
        out += v * w[k][..., None]
    return ((out + (1 << 14)) >> 15).astype(np.uint8)


def mask_blur_threshold(mask_u8):
    """cv2.GaussianBlur(m, (5,5), 0) -> cv2.threshold(127, 255, THRESH_BINARY) (dataloader.py:62-65), [P,P] uint8."""
    m = np.asarray(mask_u8).astype(np.int64)
    P = m.shape[0]
    idx = np.arange(-2, P + 2)
    idx = np.where(idx < 0, -idx, np.where(idx >= P, 2 * P - 2 - idx, idx))          # BORDER_REFLECT_101
    pad = m[idx][:, idx]
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    s = np.zeros_like(m)
    for dy in range(5):
        for dx in range(5):
            s += k[dy] * k[dx] * pad[dy:dy + P, dx:dx + P]
    v = (s + 128) >> 8
    return np.where(v > 127, 255, 0).astype(np.uint8)


def convert_cvimg_to_tensor(cvimg):
    """format.py:4-12: HWC BGR -> CHW RGB float32."""
    return np.transpose(cvimg.copy(), (2, 0, 1))[::-1, :, :].astype(np.float32)


def patch_finish(img_patch_u8, mask_patch_u8, mean, std, color_scale=(1.0, 1.0, 1.0), rm_bg=True):
    """dataloader.py:56,67-71,185-188 -> (img [3,P,P] float32, mask [1,P,P] float32)."""
    img = convert_cvimg_to_tensor(img_patch_u8)
    for c in range(3):
        img[c] = np.clip(img[c] * np.float32(color_scale[c]), 0, 255)
        if mean is not None and std is not None:
            img[c] = (img[c] - np.float32(mean[c])) / np.float32(std[c])
    mask = (mask_patch_u8[None, ...].astype(np.float32) / np.float32(255.0)).astype(np.float32)
    if rm_bg:
        img = img * mask
    return img.astype(np.float32), mask


# ------------------------------------------------------------------ geodesic.py
def compute_centroid(mask):
    """geodesic.py:4-12 (mask [1,H,W] bool) -> int16 (x, y)."""
    _, h, w = mask.shape
    grid = np.mgrid[0:h, 0:w]
    return np.array([np.sum(grid[1] * mask) / np.sum(mask), np.sum(grid[0] * mask) / np.sum(mask)]).astype(np.int16)


def fmm_distance(sources, domain, order=2):
    """Fast marching on the unit grid: distance from the `sources` pixels (value 0) through `domain` pixels (bool arrays
    [H,W]).  Pixels outside the domain (or unreachable) come back as 0, as scikit-fmm reports masked cells.

    order = 2 (scikit-fmm's default, what `skfmm.distance(m)` of geodesic.py:35,39 runs): the published scheme of its
    distance marcher - per axis the smaller FROZEN neighbour v1; if the next pixel in the same direction is frozen too and not
    larger, the one-sided second-order difference (3u - 4 v1 + v2) / 2, i.e. the quadratic term 9/4 (u - (4 v1 - v2) / 3)^2,
    else the first-order term (u - v1)^2; sum of the terms = 1, larger root.  A pixel's tentative value is RE-computed (not
    min-ed) every time one of its neighbours is frozen.  Restated from the library's documented algorithm - the package is not
    in the image (parity unpinned).  order = 1: the first-order scheme (the r03-r04 maps)."""
    H, W = domain.shape
    INF = np.inf
    u = np.full((H, W), INF)
    done = np.zeros((H, W), dtype=bool)
    heap = []
    for y, x in zip(*np.nonzero(sources & domain)):
        u[y, x] = 0.0
        heap.append((0.0, int(y), int(x)))
    heapq.heapify(heap)

    def solve1(y, x):
        a = min(u[y, x - 1] if x > 0 and domain[y, x - 1] else INF, u[y, x + 1] if x < W - 1 and domain[y, x + 1] else INF)
        b = min(u[y - 1, x] if y > 0 and domain[y - 1, x] else INF, u[y + 1, x] if y < H - 1 and domain[y + 1, x] else INF)
        lo, hi = min(a, b), max(a, b)
        if hi - lo >= 1.0:
            return lo + 1.0
        return 0.5 * (a + b + np.sqrt(2.0 - (a - b) ** 2))

    def frozen(y, x):
        return 0 <= y < H and 0 <= x < W and domain[y, x] and done[y, x]

    def solve2(y, x):
        a = b = c = 0.0
        for dy, dx in ((0, 1), (1, 0)):                        # the two axes
            v1 = v2 = INF
            for j in (-1, 1):                                  # both directions of the axis
                y1, x1 = y + j * dy, x + j * dx
                if frozen(y1, x1) and u[y1, x1] < v1:
                    v1 = u[y1, x1]
                    y2, x2 = y + 2 * j * dy, x + 2 * j * dx
                    v2 = u[y2, x2] if frozen(y2, x2) and u[y2, x2] <= v1 else INF
            if v2 < INF:
                tp = (4.0 * v1 - v2) / 3.0
                a += 2.25; b -= 2.0 * 2.25 * tp; c += 2.25 * tp * tp
            elif v1 < INF:
                a += 1.0; b -= 2.0 * v1; c += v1 * v1
        if a == 0.0:
            return INF
        c -= 1.0
        det = b * b - 4.0 * a * c
        if det < 0.0:                                          # (not reached with frozen neighbours on the unit grid; first-order fallback)
            return solve1(y, x)
        return (-b + np.sqrt(det)) / (2.0 * a)

    def relax_neighbours(y, x):
        for yy, xx in ((y, x - 1), (y, x + 1), (y - 1, x), (y + 1, x)):
            if 0 <= yy < H and 0 <= xx < W and domain[yy, xx] and not done[yy, xx]:
                if order == 2:
                    nu = solve2(yy, xx)
                    if nu != u[yy, xx]:
                        u[yy, xx] = nu
                        heapq.heappush(heap, (nu, yy, xx))
                else:
                    nu = solve1(yy, xx)
                    if nu < u[yy, xx]:
                        u[yy, xx] = nu
                        heapq.heappush(heap, (nu, yy, xx))

    if order == 2:                                             # the sources are frozen from the start (scikit-fmm: the zero level set),
        seeds, heap = heap, []                                 # the narrow band starts as their neighbours
        for _, y, x in seeds:
            done[y, x] = True
        for _, y, x in seeds:
            relax_neighbours(y, x)
    while heap:
        d, y, x = heapq.heappop(heap)
        if done[y, x] or d != u[y, x]:                         # stale entry
            continue
        done[y, x] = True
        relax_neighbours(y, x)
    u[~np.isfinite(u)] = 0.0
    return u


def compute_geodesic_dis(img, params, centers=None, order=2):
    """geodesic.py:14-54 with is_norm=True: img [1,H,W] float mask -> (weight map [1,H,W] float64, centres [n,2])."""
    mask = np.bool_(img)
    centers = compute_centroid(mask).reshape(-1, 2) if centers is None else np.asarray(centers).copy().astype(np.int16)
    for c in centers:
        if img[0, c[1], c[0]] == 0:
            return np.ones_like(img).astype(np.float16), centers
    src = np.zeros(mask.shape[1:], dtype=bool)
    for c in centers:
        src[c[1], c[0]] = True
    distance = fmm_distance(src, mask[0], order)
    distance_bg = fmm_distance(mask[0], np.ones_like(mask[0]), order)
    distance = np.exp(params[0] * (distance / np.max(distance))) + params[1]
    distance_bg = params[2] * (distance_bg / np.max(distance_bg)) + params[3]
    return (distance + distance_bg)[None, ...], centers
