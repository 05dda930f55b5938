"""SMPL linear-blend-skinning layer on one fused HIP pipeline (reference:
modules/smplpytorch/pytorch/smpl_layer.py:13-156, rodrigues_layer.py:13-52, tensutils.py:6-48).

forward(pose [B,72], betas [B,10]) -> (verts [B,6890,3], joints [B,24,3]) in metres, both minus joint
`center_idx`.  Three launches (joint regression, Rodrigues + kinematic chain in one wave, blend + skin per
vertex) replace ~100 small ATen kernels; differentiable in pose and betas (xas_smpl_lbs_bwd: three more
launches, no atomics).  The reference instantiates the layer (train.py:230-234) but never calls it on the
training path (its only call site, util.project_smpl_to_patch_kps, has no caller).

The licensed SMPL .pkl needs chumpy to unpickle (smpl_layer.py:38, serialization.py); when that import
is unavailable the constructor raises, and `SMPL_Layer.from_arrays` builds the layer from plain arrays.
Buffer names (`th_*`) follow the reference so checkpoints that contain them load.
"""
import os

import numpy as np
import torch
from torch.nn import Module

from xas_amd import ops_misc

KINTREE_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]


class SMPL_Layer(Module):
    __constants__ = ['kintree_parents', 'gender', 'center_idx', 'num_joints']

    def __init__(self, center_idx=None, gender='neutral', model_root='smpl/native/models', arrays=None):
        super().__init__()
        self.center_idx = center_idx
        self.gender = gender
        if arrays is None:
            fname = {'neutral': 'basicModel_neutral_lbs_10_207_0_v1.0.0.pkl', 'female': 'basicModel_f_lbs_10_207_0_v1.0.0.pkl',
                     'male': 'basicModel_m_lbs_10_207_0_v1.0.0.pkl'}[gender]
            self.model_path = os.path.join(model_root, fname)
            arrays = self._load_pkl(self.model_path)
        f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        self.register_buffer('th_betas', f32(arrays.get('betas', np.zeros(10))).reshape(1, -1))
        self.register_buffer('th_shapedirs', f32(arrays['shapedirs']))
        self.register_buffer('th_posedirs', f32(arrays['posedirs']))
        self.register_buffer('th_v_template', f32(arrays['v_template']).reshape(1, -1, 3))
        self.register_buffer('th_J_regressor', f32(arrays['J_regressor']))
        self.register_buffer('th_weights', f32(arrays['weights']))
        faces = arrays.get('f', np.zeros((0, 3), np.int64))
        self.register_buffer('th_faces', torch.as_tensor(np.asarray(faces).astype(np.int64)))
        parents = list(arrays.get('kintree_parents', KINTREE_PARENTS))
        parents[0] = -1                      # the pickle stores 2^32-1 for the root
        self.kintree_parents = parents
        self.num_joints = len(parents)
        self.register_buffer('_parents', torch.tensor(parents, dtype=torch.int32), persistent=False)

    @classmethod
    def from_arrays(cls, arrays, center_idx=0, gender='neutral'):
        return cls(center_idx=center_idx, gender=gender, arrays=arrays)

    @staticmethod
    def _load_pkl(path):
        import pickle
        try:
            import chumpy  # noqa: F401  (the pickle stores chumpy arrays)
        except ImportError as e:
            raise RuntimeError('loading %s needs the chumpy package; build the layer with '
                               'SMPL_Layer.from_arrays(...) from exported numpy arrays instead' % path) from e
        with open(path, 'rb') as f:
            dd = pickle.load(f, encoding='latin1')
        out = {k: np.asarray(getattr(dd[k], 'r', dd[k])) for k in ('shapedirs', 'posedirs', 'v_template', 'weights', 'f')}
        out['J_regressor'] = np.asarray(dd['J_regressor'].toarray())
        out['kintree_parents'] = [int(v) for v in dd['kintree_table'][0].tolist()]
        out['betas'] = np.zeros(out['shapedirs'].shape[-1])
        return out

    def forward(self, th_pose_axisang, th_betas=None, th_trans=None):
        B = th_pose_axisang.shape[0]
        if th_betas is None or th_betas.numel() <= 1:
            th_betas = self.th_betas.expand(B, -1)
        verts, joints = ops_misc.smpl_lbs(th_pose_axisang, th_betas, self.th_v_template, self.th_shapedirs,
                                          self.th_posedirs, self.th_J_regressor, self.th_weights, self._parents,
                                          self.center_idx if (th_trans is None or th_trans.numel() <= 1) else None)
        if th_trans is not None and th_trans.numel() > 1:
            joints = joints + th_trans.unsqueeze(1)
            verts = verts + th_trans.unsqueeze(1)
        return verts, joints
