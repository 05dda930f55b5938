"""ResNet backbone of the 3-D heat-map detector on the MI355X HIP kernels.

Mirror of the reference's modules/integral_base_modules/resnet.py:11-62 (ResNetBackbone) and of
torchvision 0.17.2's Bottleneck (v1.5, stride on the 3x3) which the reference imports at
resnet.py:2.  Parameter names equal the reference's so checkpoints load unchanged.  Each
conv -> batch-norm -> ReLU triple runs as an MFMA implicit-GEMM kernel plus fused stat / apply
kernels; the block's residual add and final ReLU are folded into bn3's apply kernel.
Stem and downsample norms are SyncBatchNorm in the reference (resnet.py:18,40); block norms
are rank-local BatchNorm2d.
"""
import torch.nn as nn

from xas_amd import layers as L
import os

from xas_amd.ops_nn import ACT_NONE, ACT_RELU, bottleneck, from_nchw

FUSED_BLOCKS = os.environ.get('XAS_FUSED_BLOCKS', '1') == '1'     # one autograd node per bottleneck (ops_nn._Bottleneck)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        k = 'kaiming_fan_out'
        self.conv1 = L.Conv2d(inplanes, planes, 1, bias=False, init=k)
        self.bn1 = L.BatchNorm2d(planes, act=ACT_RELU)
        self.conv2 = L.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False, init=k)
        self.bn2 = L.BatchNorm2d(planes, act=ACT_RELU)
        self.conv3 = L.Conv2d(planes, planes * 4, 1, bias=False, init=k)
        self.bn3 = L.BatchNorm2d(planes * 4, act=ACT_RELU)      # applied after the residual add
        self.downsample = downsample
        self.stride = stride
        self._fused_params = None

    def forward(self, x):
        if FUSED_BLOCKS and x.is_cuda:
            if self._fused_params is None:
                self._fused_params = tuple(self.parameters())
            return bottleneck(x, self)
        skip = x if self.downsample is None else self.downsample(x)
        y = self.bn1(self.conv1(x))
        y = self.bn2(self.conv2(y))
        return self.bn3(self.conv3(y), residual=skip)


class BasicBlock(nn.Module):
    """torchvision 0.17.2 BasicBlock (the reference imports it at resnet.py:2 for num_layers 18 / 34, resnet.py:5-6):
    conv3x3(stride) - BN - ReLU - conv3x3 - BN, + identity / downsample, ReLU.  Layer-by-layer on the same kernels as the
    bottleneck (the residual add and the final ReLU folded into bn2's apply kernel); no shipped YAML selects it."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        k = 'kaiming_fan_out'
        self.conv1 = L.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False, init=k)
        self.bn1 = L.BatchNorm2d(planes, act=ACT_RELU)
        self.conv2 = L.Conv2d(planes, planes, 3, padding=1, bias=False, init=k)
        self.bn2 = L.BatchNorm2d(planes, act=ACT_RELU)          # applied after the residual add
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        return self.bn2(self.conv2(self.bn1(self.conv1(x))), residual=skip)


# depth -> (block, blocks per stage, stage widths, torchvision name)  (resnet.py:5-9)
resnet_spec = {18: (BasicBlock, [2, 2, 2, 2], [64, 64, 128, 256, 512], 'resnet18'),
               34: (BasicBlock, [3, 4, 6, 3], [64, 64, 128, 256, 512], 'resnet34'),
               50: (Bottleneck, [3, 4, 6, 3], [64, 256, 512, 1024, 2048], 'resnet50'),
               101: (Bottleneck, [3, 4, 23, 3], [64, 256, 512, 1024, 2048], 'resnet101'),
               152: (Bottleneck, [3, 8, 36, 3], [64, 256, 512, 1024, 2048], 'resnet152')}


class ResNetBackbone(nn.Module):
    def __init__(self, block, layers, in_channel=3):
        super().__init__()
        self.inplanes = 64
        self.conv1 = L.Conv2d(in_channel, 64, 7, stride=2, padding=3, bias=False, init='kaiming_fan_out')
        self.bn1 = L.BatchNorm2d(64, act=ACT_RELU, sync=True)
        self.maxpool = L.MaxPool3x3s2()
        self.layer1 = self._stage(block, 64, layers[0], 1)
        self.layer2 = self._stage(block, 128, layers[1], 2)
        self.layer3 = self._stage(block, 256, layers[2], 2)
        self.layer4 = self._stage(block, 512, layers[3], 2)

    def _stage(self, block, planes, count, stride):
        proj = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            proj = nn.Sequential(
                L.Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False, init='kaiming_fan_out'),
                L.BatchNorm2d(planes * block.expansion, act=ACT_NONE, sync=True))
        blocks = [block(self.inplanes, planes, stride, proj)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes) for _ in range(1, count)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.maxpool(self.bn1(self.conv1(from_nchw(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))
