"""Oracle: D x H x W soft-argmax ("integral") heads.  TEST INFRASTRUCTURE ONLY.

Follows modules/keypoint_detector_integral_multi.py:24-88 (multi-hypothesis)
and modules/keypoint_detector_integral.py:21-65 (single hypothesis).

Layout: ``logits[b, k*D + d, h, w]`` (modules/keypoint_detector_integral_multi.py:70,74).
The reference passes (D, H, W) into parameters named (x_dim, y_dim, z_dim)
(:76 vs :36), which is only consistent because D == H == W; asserted here.

Tie rule for the depth-peak top-k (implementation-defined in torch.topk when
fewer than ``num_hypo`` strict peaks exist): larger value first, then LOWER
index first.  The HIP kernel implements the same rule.
"""
import torch


def softmax_marginals(logits, num_kp):
    """p = softmax over D*H*W per (b,k); returns px[B,K,W], py[B,K,H], pz[B,K,D].
    (keypoint_detector_integral_multi.py:69-74, 39-44)"""
    B, C, H, W = logits.shape
    D = C // num_kp
    assert D == H == W, "reference axis naming only works for D == H == W"
    p = torch.softmax(logits.reshape(B, num_kp, D * H * W), dim=2).reshape(B, num_kp, D, H, W)
    px = p.sum(dim=(2, 3))
    py = p.sum(dim=(2, 4))
    pz = p.sum(dim=(3, 4))
    return px, py, pz


def depth_peaks(pz, num_hypo):
    """Local maxima (>= both neighbours) on d in [1, D-2], value weighted,
    top ``num_hypo`` by value, returned as int64 indices in [1, D-2].
    (keypoint_detector_integral_multi.py:24-34)"""
    mid = pz[..., 1:-1]
    is_peak = (mid >= pz[..., :-2]) & (mid >= pz[..., 2:])
    score = torch.where(is_peak, mid, torch.zeros_like(mid))
    # stable descending sort == (value desc, index asc) tie rule
    order = torch.sort(score, dim=-1, descending=True, stable=True).indices
    return order[..., :num_hypo] + 1


def window_expectation(pz, idx, neighbor_size):
    """Z_h = sum_{|d'-idx|<=r} d' pz[d'] / sum pz[d']  (zero padded window).
    (keypoint_detector_integral_multi.py:55-62; the two avg_pool1d share the
    divisor because count_include_pad=True)"""
    D = pz.shape[-1]
    r = neighbor_size // 2
    d = torch.arange(D, dtype=pz.dtype)
    win = (d.view(1, 1, 1, D) - idx.unsqueeze(-1).to(pz.dtype)).abs() <= r  # [B,K,Hy,D]
    pzw = pz.unsqueeze(2) * win
    return (pzw * d).sum(-1) / pzw.sum(-1)  # [B,K,Hy]


def softargmax_multi(logits, num_kp, num_hypo, neighbor_size):
    """-> kps [B,num_hypo,K,3] fp32, depth_prob_map [K,D], z_idx [B,K,num_hypo] int64.
    (keypoint_detector_integral_multi.py:66-88)"""
    B, C, H, W = logits.shape
    D = C // num_kp
    px, py, pz = softmax_marginals(logits, num_kp)
    depth_prob_map = pz[0].clone()
    ar = torch.arange(D, dtype=logits.dtype)
    X = (px * ar).sum(-1)
    Y = (py * ar).sum(-1)
    z_idx = depth_peaks(pz, num_hypo)
    Z = window_expectation(pz, z_idx, neighbor_size)           # [B,K,Hy]
    x = X / H * 2 - 1
    y = Y / W * 2 - 1
    z = Z / D * 2 - 1
    kps = torch.stack([x.unsqueeze(1).expand(B, num_hypo, num_kp),
                       y.unsqueeze(1).expand(B, num_hypo, num_kp),
                       z.permute(0, 2, 1)], dim=-1)
    return kps.contiguous(), depth_prob_map, z_idx


def softargmax_single(logits, num_kp):
    """-> kps [B,1,K,3], depth_prob_map [K,D].  (keypoint_detector_integral.py:45-65)"""
    B, C, H, W = logits.shape
    D = C // num_kp
    px, py, pz = softmax_marginals(logits, num_kp)
    ar = torch.arange(D, dtype=logits.dtype)
    x = (px * ar).sum(-1) / H * 2 - 1
    y = (py * ar).sum(-1) / W * 2 - 1
    z = (pz * ar).sum(-1) / D * 2 - 1
    return torch.stack([x, y, z], dim=-1).unsqueeze(1), pz[0].clone()
