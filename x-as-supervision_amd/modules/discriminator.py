"""GCN pose discriminators (reference: modules/discriminator.py:8-238).

`GCNDiscriminatorDecouple` (config name 'res_sage_gcn_decouple', every shipped YAML) and `GCNSAGEDiscriminator`
run on the fused graph kernels; `GCNDiscriminator` (GCNConv with bone-length edge weights, unused by the shipped
configs) keeps its per-sample graph as a dense [B,18,18] matrix: HIP linear layers, tiny batched torch ops for
the normalisation and aggregation (autograd through the edge weights).
"""
import math

import torch
import torch.nn as nn

from modules.gcn import GCN_SAGE_residual, GCN_residual, GCN_simple
from xas_amd import layers as L


class FFNHeader(nn.Module):
    def __init__(self, in_channels, hidden_channels, p_dropout=0.2):
        super().__init__()
        self.layer1 = L.Linear(in_channels, hidden_channels)
        self.layer2 = L.Linear(hidden_channels, 1)
        self.p = p_dropout
        self.drop_mask = None          # tests inject a keep-mask to make train-mode outputs reproducible

    def forward(self, x):
        h = torch.relu(self.layer1(x))
        if self.training and self.p > 0:
            keep = self.drop_mask if self.drop_mask is not None else (torch.rand_like(h) >= self.p).to(h.dtype)
            h = h * keep / (1.0 - self.p)
        return self.layer2(h)


class GCNDiscriminator_base(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.input_dim = cfg['input_dim']
        self.hidden_dim = cfg['hidden_dim']
        self.output_dim = cfg['output_dim']
        self.disc_sup_dim = cfg['disc_sup_dim']
        self.num_nodes = cfg['num_node']
        self.use_self_loop = cfg['use_self_loop']
        self.input_layer = nn.Identity()
        self.gcn = nn.Identity()
        self.header = L.Linear(self.output_dim * self.num_nodes, 1)
        self.parent_ids, self.child_ids = None, None
        self._graph_key, self._graph = None, None
        self._pe = {}

    def cal_positional_encoding(self, keypoints):
        """PE[i,j] = sin(i / 10000^(2j/C)) for even j, cos for odd j; i = node index (discriminator.py:42-51)."""
        B, J, C = keypoints.shape
        key = (J, C, keypoints.device)
        if key not in self._pe:
            pe = torch.tensor([[math.sin(i / 10000 ** (2 * j / C)) if j % 2 == 0 else math.cos(i / 10000 ** (2 * j / C))
                                for j in range(C)] for i in range(J)], dtype=torch.float32)
            self._pe[key] = pe.to(keypoints.device)
        return self._pe[key].unsqueeze(0).expand(B, J, C)

    def graph(self, batch_size, device):
        """Constant dense mean-aggregation matrix for the skeleton set by Counter3DDisc (model.py:208-210);
        equivalent to the edge list of compute_graph_matrix (discriminator.py:53-68)."""
        key = (tuple(self.parent_ids), tuple(self.child_ids), str(device))
        if key != self._graph_key:
            n = self.num_nodes
            a = torch.eye(n) if self.use_self_loop else torch.zeros(n, n)
            a[self.parent_ids, self.child_ids] = 1.0
            a[self.child_ids, self.parent_ids] = 1.0
            a = a / a.sum(dim=1, keepdim=True).clamp_min(1.0)
            self._graph_key = key
            self._graph = (a.to(device).contiguous(), a.t().contiguous().to(device))
        return (self._graph[0], self._graph[1], batch_size, self.num_nodes)

    def _bone_index(self, device):
        """parent / child joint indices as device tensors, uploaded once (a Python-list index is uploaded - with a
        blocking copy - at every call)."""
        key = (tuple(self.parent_ids), tuple(self.child_ids), str(device))
        if getattr(self, '_bone_key', None) != key:
            self._bone_key = key
            self._bone_idx = (torch.tensor(list(self.parent_ids), dtype=torch.long, device=device),
                              torch.tensor(list(self.child_ids), dtype=torch.long, device=device))
        return self._bone_idx

    def forward_groups(self, inputs):
        """Evaluate the discriminator on several independent inputs [B,N,C] in ONE batched pass and return
        the list of logits.  Equivalent to `[self(x) for x in inputs]` (graph-LayerNorm statistics stay per
        input), but every kernel is launched once for all of them: the reference issues 28 separate
        discriminator calls per step (model.py:126-129, 241-245), ~300 tiny kernels each."""
        G = len(inputs)
        B = inputs[0].shape[0]
        out = self._forward(torch.cat(list(inputs), dim=0), groups=G)
        return list(out.split(B, dim=0))

    def forward(self, keypoints):
        return self._forward(keypoints, groups=1)

    def header_forward(self, graph_features, batch_size):
        return self.header(graph_features.reshape(batch_size, -1))


class GCNDiscriminator(GCNDiscriminator_base):
    """GCNConv discriminator with 1 / bone-length edge weights (discriminator.py:80-139; config names 'simple_gcn' /
    'res_gcn', not selected by any shipped YAML).  The per-sample graph is a dense [B, 18, 18] matrix (modules/gcn.py)."""

    def __init__(self, cfg):
        super().__init__(cfg)
        sl = self.use_self_loop
        if cfg['name'] == 'simple_gcn':
            self.name = 'SimpleGCN'
            self.gcn = nn.Sequential(GCN_simple(self.input_dim, self.hidden_dim, self_loop=sl),
                                     GCN_simple(self.input_dim, self.hidden_dim, self_loop=sl))
        elif cfg['name'] == 'res_gcn':
            self.name = 'ResGCN'
            self.num_layers = cfg['num_layers']
            self.gcn = nn.Sequential(
                GCN_simple(self.input_dim, self.hidden_dim, self_loop=sl),
                *[GCN_residual(self.hidden_dim, self.hidden_dim, self.hidden_dim, self_loop=sl, use_bn=cfg['use_bn'])
                  for _ in range(self.num_layers)],
                GCN_simple(self.hidden_dim, self.output_dim, self_loop=sl))
        else:
            raise NotImplementedError
        self.input_layer = L.Linear(self.disc_sup_dim, self.input_dim)

    def compute_graph_matrix(self, keypoints):
        """[B, N, N] weights: identity (use_self_loop) and 1 / bone length on both directions of every skeleton edge
        (discriminator.py:108-127; the reference converts this matrix to an edge list, the dense form is kept here)."""
        B = keypoints.shape[0]
        pidx, cidx = self._bone_index(keypoints.device)
        diff = keypoints[:, pidx, :] - keypoints[:, cidx, :]
        inv = 1.0 / torch.sqrt(torch.sum(diff ** 2, dim=-1))                     # [B, E]
        n = self.num_nodes
        w = torch.eye(n, device=keypoints.device).repeat(B, 1, 1) if self.use_self_loop else \
            torch.zeros(B, n, n, device=keypoints.device)
        w = w.index_put((torch.arange(B, device=keypoints.device)[:, None], pidx[None, :], cidx[None, :]), inv)
        w = w.index_put((torch.arange(B, device=keypoints.device)[:, None], cidx[None, :], pidx[None, :]), inv)
        return w

    def forward(self, keypoints):
        B = keypoints.shape[0]
        adj = self.compute_graph_matrix(keypoints)
        x = self.input_layer(keypoints.reshape(B * self.num_nodes, -1))
        return self.header_forward(self.gcn((x, adj))[0], B)

    def forward_groups(self, inputs):
        return [self(x) for x in inputs]


def _stream(hidden, out, num_layers):
    return nn.Sequential(*[GCN_SAGE_residual(hidden, hidden, hidden) for _ in range(num_layers)],
                         GCN_SAGE_residual(hidden, -1, out, single_layer=True))


class GCNSAGEDiscriminator(GCNDiscriminator_base):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.name = 'ResSAGEGCN'
        self.num_layers = cfg['num_layers']
        self.gcn = _stream(self.hidden_dim, self.output_dim, self.num_layers)
        self.use_pe = cfg['use_pe'] if 'use_pe' in cfg else False
        self.input_layer = L.Linear(self.disc_sup_dim * (2 if self.use_pe else 1), self.input_dim)

    def _forward(self, keypoints, groups=1):
        B = keypoints.shape[0]
        g = self.graph(B, keypoints.device) + (groups,)
        if self.use_pe:
            keypoints = torch.cat([keypoints, self.cal_positional_encoding(keypoints)], dim=-1)
        x = self.input_layer(keypoints.reshape(B * self.num_nodes, -1))
        return self.header_forward(self.gcn((x, g))[0], B)


class GCNDiscriminatorDecouple(GCNDiscriminator_base):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.use_pe = cfg['use_pe'] if 'use_pe' in cfg else False
        cin = self.disc_sup_dim * (2 if self.use_pe else 1)
        self.joint_input_layer = L.Linear(cin, self.input_dim)
        self.bone_input_layer = L.Linear(cin, self.input_dim)
        self.name = 'ResGCNDecouple'
        self.num_layers = cfg['num_layers']
        self.joint_gcn = _stream(self.hidden_dim, self.output_dim, self.num_layers)
        self.bone_gcn = _stream(self.hidden_dim, self.output_dim, self.num_layers)
        self.header = FFNHeader(self.output_dim * self.num_nodes * 2, 512)

    def _forward(self, keypoints, groups=1):
        B, _, dim = keypoints.shape
        pidx, cidx = self._bone_index(keypoints.device)
        bone = keypoints[:, pidx, :] - keypoints[:, cidx, :]
        bone = torch.cat([keypoints.new_zeros(B, 1, dim), bone], dim=1)       # zero row for the root node
        g = self.graph(B, keypoints.device) + (groups,)
        if self.use_pe:
            pe = self.cal_positional_encoding(keypoints)
            keypoints = torch.cat([keypoints, pe], dim=-1)
            bone = torch.cat([bone, pe], dim=-1)
        j = self.joint_gcn((self.joint_input_layer(keypoints.reshape(B * self.num_nodes, -1)), g))[0]
        b = self.bone_gcn((self.bone_input_layer(bone.reshape(B * self.num_nodes, -1)), g))[0]
        return self.header_forward(torch.cat([j, b], dim=-1), B)


# names this mirror does not replace resolve, lazily, to the reference module behind it on sys.path
from xas_amd._next import fallthrough as _fallthrough  # noqa: E402

__getattr__ = _fallthrough(__name__, __file__)
