"""Multi-hypothesis 3-D key-point detector (reference: modules/keypoint_detector_integral_multi.py:7-88).

`net` (ResNet-50 + deconv head) produces logits [B, K*D, 64, 64] with NHWC storage; the softmax over
D*H*W, the three marginals, the depth-peak top-k and the windowed expectations run as ONE fused HIP
reduction that reads the logits once (xas_head_softargmax_fwd) instead of ~15 ATen kernels and five
passes over a 604 MB tensor.  Ties between equal peak scores resolve to the lower depth bin.
"""
import torch.nn as nn

from modules.integral_base_modules.network import get_default_network_config, get_pose_net
from xas_amd import ops_head, ops_nn


class KPDetector3DMulti(nn.Module):
    def __init__(self, name, num_kp, depth_dim, num_hypo, neighbor_size, num_layers=50):
        super().__init__()
        cfg = get_default_network_config()
        cfg.depth_dim = depth_dim
        cfg.num_layers = num_layers
        self.num_hypo = num_hypo
        self.neighbor_size = neighbor_size
        self.num_kp = num_kp
        self.net = get_pose_net(cfg, num_joints=num_kp)
        self.name = name
        # the final 1x1 convolution writes the head's first-pass records from its epilogue (forward half)
        last = self.net.head.features[-1]
        if hasattr(last, 'head_kd') and depth_dim == 64:
            last.head_kd = (num_kp, depth_dim)
        self.last_peak_indices = None      # int64 [B, K, num_hypo] of the latest forward (diagnostics / tests)

    def forward(self, x):
        heatmap = self.net(x)
        kps, depth_prob_map, idx = ops_head.softargmax_multi(heatmap, self.num_kp, self.num_hypo, self.neighbor_size)
        self.last_peak_indices = idx
        return kps, depth_prob_map

    def forward_groups(self, x, groups, prefix_groups=0):
        """The reference's `groups` consecutive calls forward(x[0:B]), forward(x[B:2B]), ... as ONE pass over the
        concatenated batch: convolutions see G*B images, every batch-norm layer keeps per-call statistics and applies
        its running-statistic updates in call order (ops_nn.bn_groups).  -> kps [G*B, Hy, K, 3], depth maps [G, K, D].
        prefix_groups = P > 0 (ops_nn: prefix pass): the first P calls record no graph (the discriminator step's
        detector pass, model.py:231).  `x` is then the TAIL view [(G-P)*B, ...] of one image buffer [G*B, ...] whose
        first P*B images are those calls' inputs; -> (kps of the G-P graph calls, depth maps [G, K, D], kps of the prefix
        calls [P*B, Hy, K, 3], no graph)."""
        if not prefix_groups:
            with ops_nn.bn_groups(groups):
                heatmap = self.net(x)
            kps, depth_prob_map, idx = ops_head.softargmax_multi(heatmap, self.num_kp, self.num_hypo, self.neighbor_size,
                                                                groups=groups)
            self.last_peak_indices = idx
            return kps, depth_prob_map
        per_group = x.shape[0] // (groups - prefix_groups)
        with ops_nn.bn_groups(groups, prefix_groups, per_group):
            heatmap = self.net(x)
            kps, depth_prob_map, idx = ops_head.softargmax_multi(heatmap, self.num_kp, self.num_hypo, self.neighbor_size,
                                                                groups=groups)
        self.last_peak_indices = idx
        return kps, depth_prob_map, ops_head._last_prefix[0]
