"""Drop-in boundary: with the mirror (`x-as-supervision_amd/`) in FRONT of the reference checkout on PYTHONPATH, the
modules the mirror replaces resolve to the mirror and everything else - the dataset / loader code the real-data entry
needs (reference train_util.py:7-13,16-106; train.py:271-280; eval.py:20) - still resolves to the reference.

Runs in a child interpreter (clean sys.modules), in the build container only: it needs the reference checkout.  The
third-party packages the reference's loader imports and this image lacks (cv2, skfmm, easydict, h5py, tensorboard) are
stubbed as empty modules - only import RESOLUTION is asserted, nothing of them is called."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, 'x-as-supervision_amd')
REF = '/root/reference'

CHILD = r'''
import importlib, json, os, sys, types
PKG, REF = sys.argv[1], sys.argv[2]
sys.path[:0] = [PKG, REF]
for name in ('cv2', 'skfmm', 'h5py', 'transforms3d', 'chumpy'):
    if name not in sys.modules:
        try:
            importlib.import_module(name)
        except ImportError:
            sys.modules[name] = types.ModuleType(name)
try:
    import easydict
except ImportError:
    m = types.ModuleType('easydict')
    class EasyDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    m.EasyDict = EasyDict
    sys.modules['easydict'] = m
import matplotlib
matplotlib.use('Agg')

out = {}
def where(obj):
    mod = sys.modules[obj.__module__] if not isinstance(obj, types.ModuleType) else obj
    return os.path.realpath(mod.__file__)

from train_util import basic_data, tb_vis, pose_vis, pose_vis_3d
out['train_util.basic_data'] = where(basic_data)
out['train_util.tb_vis'] = where(tb_vis)
out['train_util.pose_vis'] = where(pose_vis)
import human_utils.dataset
out['human_utils.dataset'] = where(human_utils.dataset)
import human_utils.dataloader.dataloader as dl
out['human_utils.dataloader.dataloader'] = where(dl)
out['dataloader.compute_geodesic_dis'] = where(dl.compute_geodesic_dis)      # the loader's call resolves to the mirror's
import inspect
out['geodesic.signature'] = list(inspect.signature(dl.compute_geodesic_dis).parameters)
out['dataloader.gen_patch_image_from_box_cv'] = where(dl.gen_patch_image_from_box_cv)
import human_utils.dataloader.gpu_patch as gp
out['human_utils.dataloader.gpu_patch'] = where(gp)
import modules.base_losses.integral as integ
out['modules.base_losses.integral'] = where(integ)
import human_utils.common.visualization.pose as vp
out['human_utils.common.visualization.pose'] = where(vp)
import modules.model, modules.keypoint_detector_integral_multi, modules.keypoint_detector_integral
import modules.physique_network, modules.discriminator, modules.util, modules.base_losses.loss_func
import modules.smplpytorch.pytorch.smpl_layer as sl
import modules.integral_base_modules.network, modules.integral_base_modules.resnet, modules.integral_base_modules.deconv_head
import metrics, eval_utils
for name in ('modules.model', 'modules.keypoint_detector_integral_multi', 'modules.keypoint_detector_integral',
             'modules.physique_network', 'modules.discriminator', 'modules.util', 'modules.base_losses.loss_func',
             'modules.smplpytorch.pytorch.smpl_layer', 'modules.integral_base_modules.network',
             'modules.integral_base_modules.resnet', 'modules.integral_base_modules.deconv_head', 'metrics', 'eval_utils'):
    out[name] = where(sys.modules[name])
from modules.util import convert_world_to_patch, convert_patch_to_world        # first: reference only; second: mirrored
out['modules.util.convert_world_to_patch'] = where(convert_world_to_patch)
out['modules.util.convert_patch_to_world'] = where(convert_patch_to_world)
from eval_utils import show3Dpose, switch_points
out['eval_utils.show3Dpose'] = where(show3Dpose)
out['eval_utils.switch_points'] = where(switch_points)
import importlib.util                                                            # a package the mirror does not have at all
out['modules.smplpytorch.native.webuser.posemapper'] = os.path.realpath(      # (needs chumpy to execute: located only)
    importlib.util.find_spec('modules.smplpytorch.native.webuser.posemapper').origin)
try:
    from modules.util import no_such_name
    out['missing'] = 'resolved'
except ImportError as e:
    out['missing'] = 'ImportError'
print('RESULT ' + json.dumps(out))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='needs the reference checkout (build container only)')
def test_mirror_in_front_of_reference_resolves_both():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', PYTHONPATH='')
    r = subprocess.run([sys.executable, '-c', CHILD, PKG, REF], capture_output=True, text=True, env=env, cwd='/tmp', timeout=150)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')][-1]
    out = json.loads(line[len('RESULT '):])
    mirror = lambda p: p.startswith(os.path.realpath(PKG) + os.sep)        # noqa: E731
    ref = lambda p: p.startswith(os.path.realpath(REF) + os.sep)           # noqa: E731
    from_ref = ['train_util.basic_data', 'human_utils.dataset', 'human_utils.dataloader.dataloader',
                'dataloader.gen_patch_image_from_box_cv', 'modules.base_losses.integral',
                'human_utils.common.visualization.pose', 'modules.util.convert_world_to_patch', 'eval_utils.show3Dpose',
                'modules.smplpytorch.native.webuser.posemapper']
    from_mirror = ['train_util.tb_vis', 'train_util.pose_vis', 'dataloader.compute_geodesic_dis',
                   'human_utils.dataloader.gpu_patch', 'modules.model', 'modules.keypoint_detector_integral_multi',
                   'modules.keypoint_detector_integral', 'modules.physique_network', 'modules.discriminator', 'modules.util',
                   'modules.base_losses.loss_func', 'modules.smplpytorch.pytorch.smpl_layer',
                   'modules.integral_base_modules.network', 'modules.integral_base_modules.resnet',
                   'modules.integral_base_modules.deconv_head', 'metrics', 'eval_utils',
                   'modules.util.convert_patch_to_world', 'eval_utils.switch_points']
    for k in from_ref:
        assert ref(out[k]), (k, out[k])
    for k in from_mirror:
        assert mirror(out[k]), (k, out[k])
    # the reference's positional signature (human_utils/common/utility/geodesic.py:14)
    assert out['geodesic.signature'] == ['img', 'img_path', 'geodesic_param_list', 'centers', 'is_norm']
    assert out['missing'] == 'ImportError'


def test_mirror_alone_names_the_missing_reference():
    """Without the reference behind it, asking the mirror for a name it does not replace fails with a message that says so."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "try:\n    from train_util import basic_data\n    print('resolved')\n"
            "except ImportError as e:\n    print('ImportError', e)\n") % PKG
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd='/tmp', timeout=150,
                       env=dict(os.environ, PYTHONPATH='', PYTHONDONTWRITEBYTECODE='1'))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith('ImportError'), r.stdout
