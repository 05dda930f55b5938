"""Mask losses, graph ops for the GCN discriminator, SMPL skinning: autograd ops over the C ABI."""
import torch

from ._lib import call, ptr, query


class _MaskLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, gt, weight, use_clip):
        m, gt = m.contiguous(), gt.contiguous()
        w = weight.contiguous() if weight is not None else None
        n = m.numel()
        mode = (1 if use_clip else 0) | (2 if w is not None else 0)
        partial = torch.empty(query('xas_loss_nblk', n) * 2, device=m.device, dtype=torch.float32)
        out = torch.empty(3, device=m.device, dtype=torch.float32)
        call('xas_mask_loss_fwd', ptr(m), ptr(gt), ptr(w), n, mode, ptr(partial), ptr(out))
        ctx.save_for_backward(m, gt, w if w is not None else m.new_empty(0), out)
        ctx.mode = mode
        return out[2]

    @staticmethod
    def backward(ctx, g):
        m, gt, w, out = ctx.saved_tensors
        dm = torch.empty_like(m)
        call('xas_mask_loss_bwd', ptr(m), ptr(gt), ptr(w) if ctx.mode & 2 else None, m.numel(), ctx.mode, ptr(out),
             ptr(g.contiguous().reshape(1)), ptr(dm))
        return dm, None, None, None


def mask_loss(mask, gt, weight=None, use_clip=False):
    """Scalar value of compute_mask_reconstruction_loss(...).mean() (loss_func.py:4-16, train.py:182)."""
    return _MaskLoss.apply(mask, gt, weight, use_clip)


class _GraphAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj, adj_t, B, N):
        x = x.contiguous()
        C = x.shape[-1]
        y = torch.empty_like(x)
        call('xas_graph_aggregate', ptr(x), ptr(adj), B, N, C, ptr(y))
        ctx.save_for_backward(adj_t)
        ctx.cfg = (B, N, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        (adj_t,) = ctx.saved_tensors
        B, N, C = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call('xas_graph_aggregate', ptr(dy), ptr(adj_t), B, N, C, ptr(dx))
        return dx, None, None, None, None


def graph_aggregate(x, adj, adj_t, B, N):
    """x [B*N, C] -> mean over graph neighbours (dense row-normalised adjacency [N,N])."""
    return _GraphAggregate.apply(x, adj, adj_t, B, N)


class _GraphLayerNormRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, eps, groups):
        x = x.contiguous()
        rows_all, C = x.shape
        rows = rows_all // groups
        y = torch.empty_like(x)
        stats = torch.empty(groups * 2, device=x.device, dtype=torch.float32)
        ws = torch.empty(query('xas_gln_workspace_floats', rows * C, groups, C), device=x.device, dtype=torch.float32)
        res = residual.contiguous() if residual is not None else None
        call('xas_gln_fwd', ptr(x), ptr(gamma), ptr(beta), ptr(res), rows, C, groups, float(eps), ptr(y), ptr(stats), ptr(ws))
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.cfg = (float(eps), residual is not None, groups, rows)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        eps, has_res, groups, rows = ctx.cfg
        C = x.shape[1]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty_like(gamma)
        db = torch.empty_like(beta)
        ws = torch.empty(query('xas_gln_workspace_floats', rows * C, groups, C), device=x.device, dtype=torch.float32)
        call('xas_gln_bwd', ptr(x), ptr(beta), ptr(dy), ptr(gamma), ptr(stats), rows, C, groups, eps, ptr(dx), ptr(dg),
             ptr(db), ptr(ws))
        return dx, dg, db, (dy if has_res else None), None, None


def graph_layernorm_relu(x, gamma, beta, residual=None, eps=1e-5, groups=1):
    """relu(PyG graph-mode LayerNorm(x)) (+ residual)   (modules/gcn.py:93-110).  x: [groups*rows, C]; the
    normalisation statistics are taken per group (= per original discriminator call)."""
    return _GraphLayerNormRelu.apply(x, gamma, beta, residual, eps, groups)


def smpl_lbs(pose, betas, v_template, shapedirs, posedirs, j_regressor, weights, parents, center_idx=0):
    """SMPL forward (smpl_layer.py:63-156): -> verts [B,V,3], joints [B,24,3].  Forward only."""
    B, V = pose.shape[0], v_template.shape[-2]
    dev = pose.device
    verts = torch.empty(B, V, 3, device=dev, dtype=torch.float32)
    joints = torch.empty(B, 24, 3, device=dev, dtype=torch.float32)
    ws = torch.empty(B * (72 + 24 * 16 + 207), device=dev, dtype=torch.float32)
    call('xas_smpl_lbs_fwd', ptr(pose.contiguous()), ptr(betas.contiguous()), ptr(v_template.contiguous()),
         ptr(shapedirs.contiguous()), ptr(posedirs.contiguous()), ptr(j_regressor.contiguous()),
         ptr(weights.contiguous()), ptr(parents), B, V, -1 if center_idx is None else int(center_idx), ptr(verts),
         ptr(joints), ptr(ws))
    return verts, joints
