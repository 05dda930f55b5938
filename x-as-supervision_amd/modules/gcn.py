"""Graph building blocks of the pose discriminator (reference: modules/gcn.py:8-110).

The reference builds a sparse edge list with `nonzero()` (a host sync) and runs torch_geometric's
SAGEConv / LayerNorm.  The skeleton is fixed, so here the graph is a constant row-normalised dense
18x18 adjacency (identity + symmetric skeleton = the same neighbourhoods, self loop included), mean
aggregation is one small kernel, and PyG's graph-mode LayerNorm (normalise over the WHOLE [B*N, C]
tensor, eps added to the std) + ReLU (+ residual) is one fused kernel.  Parameter names follow PyG
(`lin_l.weight/bias`, `lin_r.weight`, `weight/bias`).
"""
import torch
import torch.nn as nn

from xas_amd import layers as L
from xas_amd import ops_misc


def my_batched_dense_to_sparse(adj):
    """Row-major non-zeros of a dense (batched) adjacency as (edge_index int64 [2,E], edge_attr [E]);
    batch b's column indices are offset by b*N (gcn.py:8-38).  Kept for API parity; the HIP path does
    not need it."""
    if adj.dim() < 2 or adj.dim() > 3:
        raise ValueError(f"Dense adjacency matrix 'adj' must be two- or three-dimensional (got {adj.dim()} dimensions)")
    if adj.dim() == 2:
        idx = adj.nonzero().t()
        return idx, adj[idx[0], idx[1]]
    B, N, M = adj.shape
    flat = adj.reshape(B * N, M)
    idx = flat.nonzero().t().clone()
    vals = flat[idx[0], idx[1]]
    idx[1] += (idx[0] // N) * M
    return idx, vals


class SAGEConv(nn.Module):
    """lin_l(mean_{j in N(i)} x_j) + lin_r(x_i); lin_l carries the bias (PyG SAGEConv, aggr='mean')."""

    def __init__(self, cin, cout):
        super().__init__()
        self.lin_l = L.Linear(cin, cout, bias=True)
        self.lin_r = L.Linear(cin, cout, bias=False)

    def forward(self, x, graph):
        adj, adj_t, B, N = graph[:4]
        return self.lin_l(ops_misc.graph_aggregate(x, adj, adj_t, B, N)) + self.lin_r(x)


class GraphLayerNorm(nn.Module):
    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class GCN_SAGE_residual(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, single_layer=False):
        super().__init__()
        self.single_layer = single_layer
        if single_layer:
            self.gc1 = SAGEConv(input_dim, output_dim)
            self.ln1 = GraphLayerNorm(output_dim)
            self.gc2 = nn.Identity()
        else:
            self.gc1 = SAGEConv(input_dim, hidden_dim)
            self.ln1 = GraphLayerNorm(hidden_dim)
            self.gc2 = SAGEConv(hidden_dim, output_dim)
            self.ln2 = GraphLayerNorm(output_dim)

    def forward(self, input):
        x, graph = input
        groups = graph[4] if len(graph) > 4 else 1          # independent calls batched along the rows
        y = ops_misc.graph_layernorm_relu(self.gc1(x, graph), self.ln1.weight, self.ln1.bias, None, self.ln1.eps, groups)
        if self.single_layer:
            return (y, graph)
        y = ops_misc.graph_layernorm_relu(self.gc2(y, graph), self.ln2.weight, self.ln2.bias, x, self.ln2.eps, groups)
        return (y, graph)
