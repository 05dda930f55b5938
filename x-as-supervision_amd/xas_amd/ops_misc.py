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


class _SmplLbs(torch.autograd.Function):
    """SMPL linear blend skinning with its backward (pose and shape gradients; the model arrays are constants)."""

    @staticmethod
    def forward(ctx, pose, betas, v_template, shapedirs, posedirs, j_regressor, weights, parents, center_idx):
        B, V = pose.shape[0], v_template.shape[-2]
        dev = pose.device
        pose, betas = pose.contiguous(), betas.contiguous()
        bufs = tuple(t.contiguous() for t in (v_template, shapedirs, posedirs, j_regressor, weights))
        verts = torch.empty(B, V, 3, device=dev, dtype=torch.float32)
        joints = torch.empty(B, 24, 3, device=dev, dtype=torch.float32)
        ws = torch.empty(B * (72 + 24 * 16 + 207), device=dev, dtype=torch.float32)
        c = -1 if center_idx is None else int(center_idx)
        call('xas_smpl_lbs_fwd', ptr(pose), ptr(betas), *(ptr(t) for t in bufs), ptr(parents), B, V, c, ptr(verts),
             ptr(joints), ptr(ws))
        ctx.save_for_backward(pose, betas, *bufs, parents, ws)
        ctx.cfg = (B, V, c)
        return verts, joints

    @staticmethod
    def backward(ctx, d_verts, d_joints):
        pose, betas, vt, sd, pd, jr, w, parents, ws = ctx.saved_tensors
        B, V, c = ctx.cfg
        d_pose, d_betas = torch.empty_like(pose), torch.empty_like(betas)
        dv = d_verts.contiguous() if d_verts is not None else torch.zeros(B, V, 3, device=pose.device)
        dj = d_joints.contiguous() if d_joints is not None else None
        ws2 = torch.empty(query('xas_smpl_lbs_bwd_workspace_floats', B, V), device=pose.device, dtype=torch.float32)
        call('xas_smpl_lbs_bwd', ptr(pose), ptr(betas), ptr(vt), ptr(sd), ptr(pd), ptr(jr), ptr(w), ptr(parents), B, V, c,
             ptr(ws), ptr(dv), ptr(dj), ptr(d_pose), ptr(d_betas), ptr(ws2))
        return d_pose, d_betas, None, None, None, None, None, None, None


def smpl_lbs(pose, betas, v_template, shapedirs, posedirs, j_regressor, weights, parents, center_idx=0):
    """SMPL layer (smpl_layer.py:63-156): -> verts [B,V,3], joints [B,24,3]; differentiable in pose and betas."""
    return _SmplLbs.apply(pose, betas, v_template, shapedirs, posedirs, j_regressor, weights, parents, center_idx)


class _PoseLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, aux, kind, w0, w1, w2):
        pred = pred.contiguous()
        B, Hy, K, _ = pred.shape
        a = aux.contiguous() if aux is not None else None
        out = torch.empty(2 + Hy, device=pred.device, dtype=torch.float32)
        call('xas_pose_loss_fwd', ptr(pred), ptr(a), B, Hy, K, kind, float(w0), float(w1), float(w2), ptr(out))
        ctx.save_for_backward(pred, a if a is not None else pred.new_empty(0), out)
        ctx.cfg = (kind, float(w0), float(w1), float(w2), aux is not None, aux is not None and aux.requires_grad and kind == 1)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pred, a, out = ctx.saved_tensors
        kind, w0, w1, w2, has_aux, aux_grad = ctx.cfg
        B, Hy, K, _ = pred.shape
        gp = torch.empty_like(pred)
        ga = torch.empty_like(pred) if aux_grad else None
        call('xas_pose_loss_bwd', ptr(pred), ptr(a) if has_aux else None, B, Hy, K, kind, w0, w1, w2, ptr(out),
             ptr(g.contiguous().reshape(1)), ptr(gp), ptr(ga))
        return gp, ga, None, None, None, None


def supervision_min(pred, gt):
    """min over hypotheses of mean((pred[:, h] - gt)^2)   (model.py:158-162 with loss_func.py:38-52)."""
    return _PoseLoss.apply(pred, gt, 0, 1.0, 0.0, 0.0)


def symmetry_min(world, w_bone, w_kp, kps=None, w_kp2d=0.0):
    """min over hypotheses of w_bone*bone_sym + w_kp*kp_sym (+ 100*w_kp2d*kp_sym(kps[..., :2]))  (model.py:104-114)."""
    return _PoseLoss.apply(world, kps, 1, w_bone, w_kp, 100.0 * w_kp2d)


class _Lsgan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        x = logits.contiguous()
        B = x.shape[0]
        Hy = x.numel() // B
        out = torch.empty(1, device=x.device, dtype=torch.float32)
        idx = torch.empty(B, device=x.device, dtype=torch.int32)
        call('xas_lsgan_fwd', ptr(x), B, Hy, float(target), ptr(out), ptr(idx))
        ctx.save_for_backward(x, idx)
        ctx.cfg = (B, Hy, float(target))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        x, idx = ctx.saved_tensors
        B, Hy, target = ctx.cfg
        gx = torch.empty_like(x)
        call('xas_lsgan_bwd', ptr(x), B, Hy, target, ptr(idx), ptr(g.contiguous().reshape(1)), ptr(gx))
        return gx, None


def lsgan_term(logits, target):
    """mean_b min_h (logits[b, h] - target)^2; logits [B, Hy, 1] or [B, 1]   (loss_func.py:54-76)."""
    return _Lsgan.apply(logits, target)
