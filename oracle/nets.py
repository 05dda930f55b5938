"""Oracle: the learnable networks (stock PyTorch CPU ops).  TEST INFRASTRUCTURE ONLY.

* ``Detector``  - ResNet-50 backbone + 3 deconv + 1x1 conv -> logits [B,K*D,64,64]
                  modules/integral_base_modules/{network.py:10-31, resnet.py:11-62,
                  deconv_head.py:3-58}.  The Bottleneck block is torchvision 0.17.2's
                  (``torchvision/models/resnet.py``, v1.5: stride on the 3x3), absent
                  from the reference tree -> restated; PARITY UNPINNED for that block.
* ``PhysiqueNet`` - modules/physique_network.py:15-59.
* ``GCNDecouple`` - modules/discriminator.py:180-238 + modules/gcn.py:79-110 with
                  torch_geometric 2.5.3 semantics restated (SAGEConv mean aggregation,
                  graph-mode LayerNorm); PARITY UNPINNED (PyG not installable here).

State-dict key names equal the reference's so the same
checkpoint loads into the oracle, the reference and the HIP modules.
SyncBatchNorm is evaluated as plain batch-norm: with no process group the
reference's SyncBatchNorm falls back to F.batch_norm as well.
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F

RESNET_DEPTHS = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
BASIC_DEPTHS = (18, 34)          # resnet.py:5-6: torchvision BasicBlock (expansion 1), the others Bottleneck (expansion 4)


class _BasicBlock(nn.Module):
    """torchvision BasicBlock restated (absent third-party code, like _Bottleneck below): conv3x3(stride)-BN-ReLU-conv3x3-BN,
    + identity / (conv1x1(stride)-BN), ReLU."""

    def __init__(self, cin, planes, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        else:
            self.downsample = None

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        skip = x if self.downsample is None else self.downsample(x)
        return F.relu(y + skip)


class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))
        else:
            self.downsample = None

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        skip = x if self.downsample is None else self.downsample(x)
        return F.relu(y + skip)


class _Backbone(nn.Module):
    def __init__(self, depth):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        basic = depth in BASIC_DEPTHS
        self.out_channels = 512 if basic else 2048
        for li, nblk in enumerate(RESNET_DEPTHS[depth]):
            planes = 64 << li
            blocks = []
            for bi in range(nblk):
                stride = 2 if (bi == 0 and li > 0) else 1
                if basic:               # resnet.py:35-36: a projection only where the stride or the width changes
                    blocks.append(_BasicBlock(cin, planes, stride, project=(stride != 1 or cin != planes)))
                    cin = planes
                else:
                    blocks.append(_Bottleneck(cin, planes, stride, project=(bi == 0)))
                    cin = planes * 4
            setattr(self, 'layer%d' % (li + 1), nn.Sequential(*blocks))

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class _DeconvHead(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        mods = []
        for i in range(3):
            mods += [nn.ConvTranspose2d(cin if i == 0 else 256, 256, 4, 2, 1, bias=False),
                     nn.BatchNorm2d(256), nn.ReLU()]
        mods.append(nn.Conv2d(256, cout, 1, bias=True))
        self.features = nn.ModuleList(mods)

    def forward(self, x):
        for m in self.features:
            x = m(x)
        return x


class _PoseNet(nn.Module):
    def __init__(self, depth, cout):
        super().__init__()
        self.backbone = _Backbone(depth)
        self.head = _DeconvHead(self.backbone.out_channels, cout)

    def forward(self, x):
        return self.head(self.backbone(x))


def kaiming_init_(module, generator=None):
    """resnet.py:26-32 / deconv_head.py:42-53: kaiming_normal(fan_out, relu) on every
    Conv2d / ConvTranspose2d weight, zero bias, BN weight 1 / bias 0."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            w = m.weight
            # torch's fan_out = size(0) * receptive field for both layer types
            fan_out = w.shape[0] * w.shape[2] * w.shape[3]
            with torch.no_grad():
                w.normal_(0.0, math.sqrt(2.0 / fan_out), generator=generator)
                if m.bias is not None:
                    m.bias.zero_()


class Detector(nn.Module):
    """KPDetector3DMulti / KPDetector3D minus the soft-argmax head: ``net(x)``."""

    def __init__(self, num_kp, depth_dim, num_layers=50):
        super().__init__()
        self.num_kp = num_kp
        self.net = _PoseNet(num_layers, num_kp * depth_dim)
        kaiming_init_(self.net)

    def forward(self, x):
        return self.net(x)


class PhysiqueNet(nn.Module):
    """modules/physique_network.py:15-59; conv biases present, LeakyReLU(0.01),
    bilinear x2 with align_corners=False, final sigmoid."""

    def __init__(self, feats, num_parts=1):
        super().__init__()

        def block(cin, cout, stride=1, up=False):
            mods = [nn.Upsample(scale_factor=2, mode='bilinear')] if up else []
            mods += [nn.Conv2d(cin, cout, 3, stride, 1), nn.BatchNorm2d(cout), nn.LeakyReLU()]
            return nn.Sequential(*mods)

        enc = [block(num_parts, feats[0])]
        for i in range(1, len(feats)):
            enc += [block(feats[i - 1], feats[i - 1]), block(feats[i - 1], feats[i], stride=2)]
        dec = []
        for i in range(len(feats) - 1, 0, -1):
            dec += [block(feats[i], feats[i]), block(feats[i], feats[i - 1], up=True)]
        dec.append(nn.Conv2d(feats[0], 1, 3, 1, 1))
        self.encoder = nn.Sequential(*enc)
        self.decoder = nn.Sequential(*dec)

    def forward(self, x):
        return torch.sigmoid(self.decoder(self.encoder(x)))


# ---------------------------------------------------------------- GCN discriminator
class _Lin(nn.Module):
    def __init__(self, cin, cout, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class _Sage(nn.Module):
    """PyG SAGEConv(aggr='mean'): lin_l(mean_{j in N(i)} x_j) + lin_r(x_i);
    lin_l has the bias, lin_r has none."""

    def __init__(self, cin, cout):
        super().__init__()
        self.lin_l = _Lin(cin, cout, True)
        self.lin_r = _Lin(cin, cout, False)

    def forward(self, x, adj_mean):
        # x [B,N,C]; adj_mean [N,N] row-normalised (I + skeleton) adjacency
        return self.lin_l(torch.einsum('ij,bjc->bic', adj_mean, x)) + self.lin_r(x)


class _GraphLN(nn.Module):
    """PyG norm.LayerNorm(mode='graph') with batch=None: normalise over the WHOLE
    [B*N, C] tensor, eps added to the std, then per-channel affine."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))

    def forward(self, x):
        x = x - x.mean()
        return x / (x.std(unbiased=False) + 1e-5) * self.weight + self.bias


class _SageRes(nn.Module):
    def __init__(self, c, single):
        super().__init__()
        self.single = single
        self.gc1 = _Sage(c, c)
        self.ln1 = _GraphLN(c)
        if not single:
            self.gc2 = _Sage(c, c)
            self.ln2 = _GraphLN(c)

    def forward(self, x, adj):
        y = F.relu(self.ln1(self.gc1(x, adj)))
        if self.single:
            return y
        y = F.relu(self.ln2(self.gc2(y, adj)))
        return x + y


class _Header(nn.Module):
    def __init__(self, cin, hidden):
        super().__init__()
        self.layer1 = nn.Linear(cin, hidden)
        self.layer2 = nn.Linear(hidden, 1)

    def forward(self, x, drop_mask=None):
        h = F.relu(self.layer1(x))
        if drop_mask is not None:       # train-mode Dropout(0.2) with an injected keep mask
            h = h * drop_mask / 0.8
        return self.layer2(h)


def positional_encoding(num_nodes, dim):
    """discriminator.py:42-51."""
    pe = torch.zeros(num_nodes, dim)
    for i in range(num_nodes):
        for j in range(dim):
            a = i / 10000 ** (2 * j / dim)
            pe[i, j] = math.sin(a) if j % 2 == 0 else math.cos(a)
    return pe


def mean_adjacency(num_nodes, parents, children, self_loop=True):
    """Row-normalised dense form of the edge list built at discriminator.py:53-68."""
    a = torch.eye(num_nodes) if self_loop else torch.zeros(num_nodes, num_nodes)
    a[parents, children] = 1.0
    a[children, parents] = 1.0
    return a / a.sum(dim=1, keepdim=True)


def batched_dense_to_sparse(adj):
    """modules/gcn.py:8-38 (3-D branch): row-major non-zeros of the flattened
    [B*N, N] matrix, column index offset by b*N."""
    B, N, M = adj.shape
    nz = adj.reshape(B * N, M).nonzero()
    rows, cols = nz[:, 0], nz[:, 1]
    vals = adj.reshape(B * N, M)[rows, cols]
    cols = cols + (rows // N) * M
    return torch.stack([rows, cols]), vals


class GCNDecouple(nn.Module):
    """GCNDiscriminatorDecouple, discriminator.py:180-238."""

    def __init__(self, cfg):
        super().__init__()
        self.name = 'ResGCNDecouple'
        c = cfg['hidden_dim']
        self.num_nodes = cfg['num_node']
        self.use_pe = cfg.get('use_pe', False)
        self.use_self_loop = cfg['use_self_loop']
        cin = cfg['disc_sup_dim'] * (2 if self.use_pe else 1)
        self.joint_input_layer = nn.Linear(cin, cfg['input_dim'])
        self.bone_input_layer = nn.Linear(cin, cfg['input_dim'])
        nl = cfg['num_layers']
        self.joint_gcn = nn.ModuleList([_SageRes(c, False) for _ in range(nl)] + [_SageRes(c, True)])
        self.bone_gcn = nn.ModuleList([_SageRes(c, False) for _ in range(nl)] + [_SageRes(c, True)])
        self.header = _Header(cfg['output_dim'] * self.num_nodes * 2, 512)
        self.parent_ids = self.child_ids = None

    def forward(self, kp, drop_mask=None):
        B, N, C = kp.shape
        bone = kp[:, self.parent_ids] - kp[:, self.child_ids]
        bone = torch.cat([torch.zeros(B, 1, C, dtype=kp.dtype), bone], dim=1)
        adj = mean_adjacency(N, self.parent_ids, self.child_ids, self.use_self_loop)
        if self.use_pe:
            pe = positional_encoding(N, C).unsqueeze(0).expand(B, N, C)
            kp = torch.cat([kp, pe], -1)
            bone = torch.cat([bone, pe], -1)
        j = self.joint_input_layer(kp)
        for blk in self.joint_gcn:
            j = blk(j, adj)
        b = self.bone_input_layer(bone)
        for blk in self.bone_gcn:
            b = blk(b, adj)
        return self.header(torch.cat([j, b], -1).reshape(B, -1), drop_mask)


# ---------------------------------------------------------------- GCNConv discriminator (no shipped config selects it)
class _GCNConv(nn.Module):
    """torch_geometric 2.5.3 GCNConv(in, out, add_self_loops) on an edge list (published algorithm, gcn_conv.py:
    gcn_norm + propagate): self loops of weight 1 are added to the nodes that have none (existing ones keep their weight),
    deg_i = sum of the weights of the edges ending in i, norm_e = deg^-1/2[source] * w_e * deg^-1/2[target] (1/sqrt(0) -> 0),
    out_i = sum over edges (j -> i) of norm_e * (x W)_j, plus the bias (the linear layer itself has none)."""

    def __init__(self, cin, cout, add_self_loops=True):
        super().__init__()
        self.lin = _Lin(cin, cout, False)
        self.bias = nn.Parameter(torch.zeros(cout))
        self.add_self_loops = add_self_loops

    def forward(self, x, edge_index, edge_weight):
        n = x.shape[0]
        row, col, w = edge_index[0], edge_index[1], edge_weight
        if self.add_self_loops:
            loop = row == col
            lw = torch.ones(n, dtype=x.dtype).index_put((row[loop],), w[loop])
            ar = torch.arange(n)
            row, col, w = torch.cat([row[~loop], ar]), torch.cat([col[~loop], ar]), torch.cat([w[~loop], lw])
        deg = torch.zeros(n, dtype=x.dtype).index_add(0, col, w)
        dis = deg.pow(-0.5)
        dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
        norm = dis[row] * w * dis[col]
        xw = self.lin(x)
        return torch.zeros_like(xw).index_add(0, col, norm[:, None] * xw[row]) + self.bias


class _GCNSimple(nn.Module):
    """gcn.py:40-49."""

    def __init__(self, cin, cout, self_loop):
        super().__init__()
        self.gc = _GCNConv(cin, cout, self_loop)

    def forward(self, x, ei, ew):
        return F.relu(self.gc(x, ei, ew))


class _GCNResidual(nn.Module):
    """gcn.py:52-77: gc1 -> (bn) -> relu -> dropout -> gc2 -> (bn) -> relu -> dropout, + input; ONE norm module for both."""

    def __init__(self, c, self_loop, use_bn, p_dropout=0.5):
        super().__init__()
        self.gc1 = _GCNConv(c, c, self_loop)
        self.gc2 = _GCNConv(c, c, self_loop)
        self.use_bn = use_bn
        if use_bn:
            self.bn = nn.BatchNorm1d(c)          # nn.SyncBatchNorm without a process group = plain batch norm
        self.p = p_dropout

    def forward(self, x, ei, ew):
        res = x
        for gc in (self.gc1, self.gc2):
            x = gc(x, ei, ew)
            if self.use_bn:
                x = self.bn(x)
            x = F.dropout(F.relu(x), self.p, self.training)
        return x + res


class GCNConvDisc(nn.Module):
    """GCNDiscriminator, discriminator.py:80-139 ('simple_gcn' / 'res_gcn'): edge weights 1 / bone length."""

    def __init__(self, cfg):
        super().__init__()
        self.num_nodes, self.use_self_loop = cfg['num_node'], cfg['use_self_loop']
        i, h, o, sl = cfg['input_dim'], cfg['hidden_dim'], cfg['output_dim'], cfg['use_self_loop']
        if cfg['name'] == 'simple_gcn':
            self.name = 'SimpleGCN'
            self.gcn = nn.ModuleList([_GCNSimple(i, h, sl), _GCNSimple(i, h, sl)])
        else:
            self.name = 'ResGCN'
            self.gcn = nn.ModuleList([_GCNSimple(i, h, sl)] + [_GCNResidual(h, sl, cfg['use_bn']) for _ in range(cfg['num_layers'])]
                                     + [_GCNSimple(h, o, sl)])
        self.input_layer = nn.Linear(cfg['disc_sup_dim'], i)
        self.header = nn.Linear(o * self.num_nodes, 1)
        self.parent_ids = self.child_ids = None

    def forward(self, kp):
        B, N, _ = kp.shape
        diff = kp[:, self.parent_ids] - kp[:, self.child_ids]
        inv = 1.0 / torch.sqrt((diff ** 2).sum(-1))
        w = torch.eye(N, dtype=kp.dtype).repeat(B, 1, 1) if self.use_self_loop else torch.zeros(B, N, N, dtype=kp.dtype)
        bi = torch.arange(B)[:, None]
        w = w.index_put((bi, torch.tensor(self.parent_ids)[None], torch.tensor(self.child_ids)[None]), inv)
        w = w.index_put((bi, torch.tensor(self.child_ids)[None], torch.tensor(self.parent_ids)[None]), inv)
        ei, ew = batched_dense_to_sparse(w)
        x = self.input_layer(kp).reshape(B * N, -1)
        for blk in self.gcn:
            x = blk(x, ei, ew)
        return self.header(x.reshape(B, -1))
