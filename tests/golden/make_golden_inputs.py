"""Input dictionaries shared by tests/golden/make_golden.py (g_tbvis) and tests/test_tbvis.py: batch / output key sets of
dataloader.py:166-191 and model.py:68-96,153-173,238-248 filled with random tensors."""
import torch


def tbvis_inputs():
    """A batch dict / output dict with the key sets of dataloader.py:166-191 and model.py:68-96,153-173,238-248."""
    cams = [0, 1]
    x, out = {'act': ['act_02_subact_01'], 'cam_0_img_path': ['s_01_act_02_subact_01_ca_01/frame.jpg']}, {}
    for c in cams:
        k = 'cam_%d' % c
        x[k + '_img'] = torch.rand(2, 3, 32, 32)
        x[k + '_mask'] = (torch.rand(2, 1, 32, 32) > 0.5).float()
        x[k + '_joints'] = torch.rand(2, 18, 3) * 31
        x[k + '_geodesic_dis'] = 1 + torch.rand(2, 1, 32, 32)
        x[k + '_geodesic_center'] = torch.tensor([[[16, 16]], [[15, 17]]])
        x[k + '_pseudo_img'] = torch.rand(2, 3, 32, 32)
        x[k + '_pseudo_joints'] = torch.rand(2, 18, 3)
        x[k + '_k_mat'] = torch.eye(3).repeat(2, 1, 1)
        x[k + '_img_path'] = x.get(k + '_img_path', ['p'])
        out['pose_2d_pred_%s_ori' % k] = torch.rand(1, 18, 3) * 2 - 1
        out['depth_map_' + k] = torch.rand(18, 64)
        out['pose_3d_depth_' + k] = torch.rand(2, 18, 3) * 1000
        out['mask_heatmap_line_' + k] = torch.rand(2, 1, 32, 32)
        out['mask_physique_' + k] = torch.rand(1, 1, 32, 32)
        out['pose_2d_pred_%s_pseudo' % k] = torch.rand(1, 18, 3) * 2 - 1
        out['pose_3d_pred_%s_pseudo' % k] = torch.rand(1, 18, 3)
        out['pose_3d_gt_%s_pseudo' % k] = torch.rand(1, 18, 3)
        out['pose_smpl_2d_' + k] = torch.rand(1, 18, 3)
        out['pose_smpl_3d_' + k] = torch.rand(1, 18, 3)
        out['smpl_logits_' + k] = torch.rand(1, 1)
        out['pred_logits_' + k] = torch.rand(1, 1)
    out['kp_gt_world'] = torch.rand(1, 18, 3)
    losses = {'symmetry': torch.rand(()), 'smpl_gen': torch.rand(()), 'reconstruction': torch.rand(2, 1, 4, 4)}
    return x, out, losses
