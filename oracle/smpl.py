"""Oracle: SMPL linear-blend skinning.  TEST INFRASTRUCTURE ONLY.

Follows modules/smplpytorch/pytorch/smpl_layer.py:63-156,
rodrigues_layer.py:13-52, tensutils.py:6-48 and modules/util.py:331-341.
"""
import torch

SMPL_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)


def rodrigues(axisang):
    """[N,3] -> [N,3,3] via unit quaternion; angle = ||axisang + 1e-8||
    (rodrigues_layer.py:41-52, 13-38)."""
    angle = (axisang + 1e-8).norm(dim=1, keepdim=True)
    axis = axisang / angle
    half = angle * 0.5
    q = torch.cat([torch.cos(half), torch.sin(half) * axis], dim=1)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([
        w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
        2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
        2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z], dim=1).view(-1, 3, 3)


def smpl_lbs(pose, betas, v_template, shapedirs, posedirs, j_regressor, weights,
             parents=SMPL_PARENTS, center_idx=0):
    """pose [B,72], betas [B,10] (non-zero), buffers as registered at smpl_layer.py:40-55
    (v_template [1,V,3], shapedirs [V,3,10], posedirs [V,3,207], J_regressor [24,V],
    weights [V,24]) -> verts [B,V,3], joints [B,24,3], both minus joint ``center_idx``."""
    B = pose.shape[0]
    J = len(parents)
    R = rodrigues(pose.reshape(B * J, 3)).view(B, J, 3, 3)
    eye = torch.eye(3, dtype=pose.dtype)
    pose_map = (R[:, 1:] - eye).reshape(B, (J - 1) * 9)
    v_shaped = v_template + torch.einsum('vcs,bs->bvc', shapedirs, betas)
    joints0 = torch.einsum('jv,bvc->bjc', j_regressor, v_shaped)
    v_posed = v_shaped + torch.einsum('vcp,bp->bvc', posedirs, pose_map)

    def rigid(rot, t):
        top = torch.cat([rot, t.unsqueeze(-1)], dim=2)
        bot = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=pose.dtype).view(1, 1, 4).expand(B, 1, 4)
        return torch.cat([top, bot], dim=1)

    G = [rigid(R[:, 0], joints0[:, 0])]
    for i in range(1, J):
        G.append(G[parents[i]] @ rigid(R[:, i], joints0[:, i] - joints0[:, parents[i]]))
    G = torch.stack(G, dim=1)                                   # [B,J,4,4]
    jh = torch.cat([joints0, torch.zeros(B, J, 1, dtype=pose.dtype)], dim=2)
    corr = torch.einsum('bjrc,bjc->bjr', G, jh)                 # G_i (J_i, 0)
    G2 = G.clone()
    G2[..., 3] = G2[..., 3] - corr
    T = torch.einsum('vj,bjrc->bvrc', weights, G2)              # [B,V,4,4]
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=pose.dtype)], dim=2)
    verts = torch.einsum('bvrc,bvc->bvr', T, vh)[..., :3]
    jtr = G[:, :, :3, 3]
    if center_idx is not None:
        c = jtr[:, center_idx:center_idx + 1]
        jtr = jtr - c
        verts = verts - c
    return verts, jtr


def smpl_to_h36m(verts, h36m_regressor):
    """util.py:331-341: regress 17 joints, swap L/R arms, append thorax, root-centre."""
    j = torch.einsum('bki,lk->bli', verts, h36m_regressor)
    j = j[:, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14, 15, 16, 11, 12, 13]]
    j = torch.cat([j, j[:, [11, 14]].mean(dim=1, keepdim=True)], dim=1)
    return j - j[:, [0]]
