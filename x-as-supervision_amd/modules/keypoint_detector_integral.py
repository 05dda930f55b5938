"""Single-hypothesis 3-D key-point detector (reference: modules/keypoint_detector_integral.py:7-65):
softmax over D*H*W and the plain three-axis expectation, fused into one HIP reduction."""
import torch.nn as nn

from modules.integral_base_modules.network import get_default_network_config, get_pose_net
from xas_amd import ops_head, ops_nn


class KPDetector3D(nn.Module):
    def __init__(self, name, num_kp, depth_dim, num_layers=50):
        super().__init__()
        cfg = get_default_network_config()
        cfg.depth_dim = depth_dim
        cfg.num_layers = num_layers
        self.num_kp = num_kp
        self.net = get_pose_net(cfg, num_joints=num_kp)
        self.name = name

    def forward(self, x):
        kps, depth_prob_map = ops_head.softargmax_single(self.net(x), self.num_kp)
        return kps, depth_prob_map          # kps [B, 1, num_kp, 3], aligned with the multi-hypothesis layout

    def forward_groups(self, x, groups):
        """`groups` consecutive reference calls as one camera-batched pass (see KPDetector3DMulti.forward_groups)."""
        with ops_nn.bn_groups(groups):
            heatmap = self.net(x)
        return ops_head.softargmax_single(heatmap, self.num_kp, groups=groups)
