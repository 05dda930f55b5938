"""ResPoseNet = ResNet backbone + deconvolution head (reference: integral_base_modules/network.py:10-53).

`get_default_network_config()` returns an attribute dict with the reference's defaults
(network.py:33-44); no easydict dependency.  ImageNet initialisation (network.py:46-53) needs
torchvision's model zoo and the network: when torchvision is importable it is used, otherwise the
seeded Kaiming initialisation stays (what the benchmark and the tests use).
"""
import torch.nn as nn

from .deconv_head import DeconvHead
from .resnet import ResNetBackbone, resnet_spec


class _Cfg(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class ResPoseNet(nn.Module):
    def __init__(self, backbone, head):
        super().__init__()
        self.backbone = backbone
        self.head = head
        self._counters = None

    def forward(self, x):
        if self.training:
            if self._counters is None:
                from xas_amd.layers import SharedBatchCounters
                self._counters = SharedBatchCounters(self)
            self._counters.bump()
        return self.head(self.backbone(x))


def get_default_network_config():
    return _Cfg(from_model_zoo=True, pretrained='', num_layers=50, num_deconv_layers=3, num_deconv_filters=256,
                num_deconv_kernel=4, final_conv_kernel=1, depth_dim=1, input_channel=3)


def init_pose_net(pose_net, cfg):
    if cfg.from_model_zoo:
        try:
            import torchvision.models as models
        except ImportError:
            return pose_net                      # offline / no torchvision: keep the Kaiming init
        name = resnet_spec[cfg.num_layers][3]
        try:
            zoo = getattr(models, name)(weights='DEFAULT').state_dict()
        except Exception:
            return pose_net
        zoo.pop('fc.weight', None)
        zoo.pop('fc.bias', None)
        pose_net.backbone.load_state_dict(zoo)
    return pose_net


def get_pose_net(cfg, num_joints):
    block, layers, channels, _ = resnet_spec[cfg.num_layers]
    backbone = ResNetBackbone(block, layers, cfg.input_channel)
    head = DeconvHead(channels[-1], cfg.num_deconv_layers, cfg.num_deconv_filters, cfg.num_deconv_kernel,
                      cfg.final_conv_kernel, num_joints, cfg.depth_dim)
    return init_pose_net(ResPoseNet(backbone, head), cfg)


# names this mirror does not replace resolve, lazily, to the reference module behind it on sys.path
from xas_amd._next import fallthrough as _fallthrough  # noqa: E402

__getattr__ = _fallthrough(__name__, __file__)
