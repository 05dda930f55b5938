"""Physique mask generator: conv encoder/decoder [32,64,128] (reference: modules/physique_network.py:4-59).

Every `C` = 3x3 conv (with bias) + SyncBatchNorm + LeakyReLU(0.01); down = stride 2; up = bilinear x2
(align_corners=False) in front of the conv; last 3x3 conv to one channel + sigmoid.  The 32..128-channel
convs run on the MFMA implicit-GEMM kernels; the 1-channel input / output convs use the direct kernels.
Sequential indices match the reference so `encoder.N.M` / `decoder.N.M` checkpoint keys load unchanged.
"""
import torch.nn as nn

from xas_amd import layers as L
from xas_amd import ops_nn as F
from xas_amd.ops_nn import ACT_LEAKY


class PhysiqueMaskGenerator(nn.Module):
    def __init__(self, num_features, num_parts=1):
        super().__init__()
        self.num_features = num_features
        self.num_parts = num_parts
        self.encoder, self.decoder = self.define_network(num_features)

    @staticmethod
    def _unit(cin, cout, stride=1, up=False):
        mods = [L.Upsample2x()] if up else []
        mods += [L.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=True),
                 L.BatchNorm2d(cout, act=ACT_LEAKY, sync=True), nn.Identity()]
        return nn.Sequential(*mods)

    def define_network(self, f):
        enc = [self._unit(self.num_parts, f[0])]
        for i in range(1, len(f)):
            enc += [self._unit(f[i - 1], f[i - 1]), self._unit(f[i - 1], f[i], stride=2)]
        dec = []
        for i in range(len(f) - 1, 0, -1):
            dec += [self._unit(f[i], f[i]), self._unit(f[i], f[i - 1], up=True)]
        dec.append(L.Conv2d(f[0], 1, 3, stride=1, padding=1, bias=True))
        return nn.Sequential(*enc), nn.Sequential(*dec)

    def forward(self, input):
        return F.sigmoid(self.decoder(self.encoder(input)))

    def forward_groups(self, input, groups):
        """`groups` consecutive reference calls on equal sub-batches as ONE pass (per-call batch-norm statistics and
        running-statistic updates in call order: ops_nn.bn_groups)."""
        with F.bn_groups(groups):
            return F.sigmoid(self.decoder(self.encoder(input)))
