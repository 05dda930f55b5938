"""Deconvolution head: 3 x (ConvTranspose2d k4 s2 p1 -> SyncBN -> ReLU) + 1x1 conv with bias.

Mirror of the reference's modules/integral_base_modules/deconv_head.py:3-58.  The transposed
convolutions run as the stride-phase data-gradient MFMA kernel (4 sub-pixel phases, 2x2 taps each),
the final 256 -> num_joints*depth_dim projection as a plain MFMA GEMM over NHWC pixels.
`features` keeps the reference's indices (0,1,3,4,6,7,9 hold parameters; 2,5,8 are the ReLUs,
which here are fused into the preceding norm).
"""
import torch.nn as nn

from xas_amd import layers as L
from xas_amd.ops_nn import ACT_RELU


class DeconvHead(nn.Module):
    def __init__(self, in_channels, num_layers, num_filters, kernel_size, conv_kernel_size, num_joints, depth_dim,
                 with_bias_end=True):
        super().__init__()
        if kernel_size != 4:
            raise NotImplementedError('only the shipped 4x4 stride-2 deconvolution is built (network.py:38)')
        if conv_kernel_size not in (1, 3):
            raise ValueError('Only support kenerl 1 and 3')
        feats = []
        for i in range(num_layers):
            feats += [L.ConvTranspose2d(in_channels if i == 0 else num_filters, num_filters, 4, 2, 1),
                      L.BatchNorm2d(num_filters, act=ACT_RELU, sync=True), nn.Identity()]
        pad = 0 if conv_kernel_size == 1 else 1
        feats.append(L.Conv2d(num_filters, num_joints * depth_dim, conv_kernel_size, padding=pad,
                              bias=with_bias_end, init='kaiming_fan_out'))
        if with_bias_end:
            nn.init.zeros_(feats[-1].bias)
        else:
            feats += [L.BatchNorm2d(num_joints * depth_dim, act=ACT_RELU, sync=True), nn.Identity()]
        self.features = nn.ModuleList(feats)

    def forward(self, x):
        for layer in self.features:
            x = layer(x)
        return x
