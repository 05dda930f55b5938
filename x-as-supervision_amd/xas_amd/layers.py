"""nn.Module wrappers with torch-compatible parameter names (weight, bias, running_mean, ...), so
reference checkpoints load unchanged, running on the HIP ops."""
import math

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops_nn as F


def _kaiming_fan_out_(w):
    """nn.init.kaiming_normal_(mode='fan_out', nonlinearity='relu') (resnet.py:28, deconv_head.py:45,53)."""
    fan_out = w.shape[0] * w[0][0].numel()
    with torch.no_grad():
        w.normal_(0.0, math.sqrt(2.0 / fan_out))


class Conv2d(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True, init='default'):
        super().__init__()
        self.stride, self.padding = stride, padding
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        if init == 'kaiming_fan_out':
            _kaiming_fan_out_(self.weight)
        else:                                   # torch.nn.Conv2d default
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
            if bias:
                bound = 1 / math.sqrt(cin * k * k)
                nn.init.uniform_(self.bias, -bound, bound)
        self._cache = F._PackCache()
        self.head_kd = None            # (joints, depth bins): this conv produces the logits of the soft-argmax head - its epilogue
                                       # then also emits the head's first-pass records (ops_nn.conv2d, xas_conv_fwd_head)

    def forward(self, x):
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self._cache, self.head_kd)


class ConvTranspose2d(nn.Module):
    """weight layout [Cin, Cout, k, k] as torch.nn.ConvTranspose2d; no bias (deconv_head.py:27-29)."""

    def __init__(self, cin, cout, k, stride, padding):
        super().__init__()
        self.stride, self.padding = stride, padding
        self.weight = nn.Parameter(torch.empty(cin, cout, k, k))
        _kaiming_fan_out_(self.weight)
        self._cache = F._PackCache()

    def forward(self, x):
        return F.conv_transpose2d(x, self.weight, self.stride, self.padding, self._cache)


class Linear(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(cin)
            nn.init.uniform_(self.bias, -bound, bound)
        self._cache = F._PackCache()

    def forward(self, x):
        return F.linear(x, self.weight, self.bias, self._cache)


class BatchNorm2d(nn.Module):
    """Training-mode batch norm fused with the following activation (and residual add).
    sync=True reproduces nn.SyncBatchNorm: statistics are exchanged across the default process
    group when one is initialised with world_size > 1 (resnet.py:18,40; deconv_head.py:30;
    physique_network.py:18,25,33); otherwise it is rank-local like torchvision's BatchNorm2d."""

    def __init__(self, c, act=F.ACT_NONE, sync=False, eps=1e-5, momentum=0.1):
        super().__init__()
        self.act, self.sync, self.eps, self.momentum = act, sync, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self._nbt_shared = False       # True once a parent owns the counter (share_batch_counters)

    def sync_group(self):
        """Process group of the statistic exchange, or None for a rank-local norm."""
        if self.sync and self.training:
            from .dp import dp_active
            if dp_active():
                return dist.group.WORLD
        return None

    def count_batch(self):
        """num_batches_tracked += number of camera groups in this call (1 for a plain call) unless a parent owns the
        counter (SharedBatchCounters): a camera-batched pass stands for that many reference calls."""
        if self.training and not self._nbt_shared:
            from . import streams
            n = F.current_groups()
            if streams.forked():
                with torch.cuda.stream(streams.book_stream()):
                    self.num_batches_tracked += n
            else:
                self.num_batches_tracked += n

    def forward(self, x, residual=None):
        group = self.sync_group()
        self.count_batch()
        return F.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, residual, self.training,
                            self.momentum, self.eps, self.act, group)


class MaxPool3x3s2(nn.Module):
    def forward(self, x):
        return F.maxpool3x3s2(x)


class Upsample2x(nn.Module):
    def forward(self, x):
        return F.upsample2x(x)


class SharedBatchCounters:
    """All `num_batches_tracked` buffers of a network that runs every norm exactly once per forward are
    re-homed into one int64 arena, so a forward bumps them with ONE launch instead of one per layer (56 in the
    detector).  The buffers stay ordinary state-dict entries (views of the arena)."""

    def __init__(self, net):
        self.net = net
        self.arena = None

    def _bind(self):
        mods = [m for m in self.net.modules() if isinstance(m, BatchNorm2d)]
        dev = mods[0].num_batches_tracked.device
        arena = torch.stack([m.num_batches_tracked.reshape(()) for m in mods]).to(dev)
        for i, m in enumerate(mods):
            m._buffers['num_batches_tracked'] = arena[i]
            m._nbt_shared = True
        self.arena, self.mods = arena, mods

    def bump(self):
        ok = self.arena is not None and all(
            m._buffers['num_batches_tracked'].data_ptr() == self.arena.data_ptr() + 8 * i for i, m in enumerate(self.mods))
        if not ok:                      # first use, or the buffers were replaced (.to(), load_state_dict)
            self._bind()
        from . import streams
        n = F.current_groups()          # a camera-batched pass counts once per camera
        if streams.forked():            # concurrent camera streams: serialise the read-modify-write
            with torch.cuda.stream(streams.book_stream()):
                self.arena += n
        else:
            self.arena += n
