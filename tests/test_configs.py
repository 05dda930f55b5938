"""The configuration builder of the synthetic harness (xas_amd.synthetic.model_config) must equal the shipped YAML
files field for field.  tests/golden/configs.json is a data fixture written by tests/golden/make_golden.py (g_configs)
from /root/reference/config/*.yaml: model_params, train_params, the camera list and the dataset name of each file."""
import json
import os

import pytest

from xas_amd.synthetic import CONFIG_NAMES, model_config

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, 'golden', 'configs.json')) as f:
    FIXTURE = json.load(f)


def test_every_yaml_is_known():
    assert sorted(FIXTURE) == sorted(CONFIG_NAMES)


@pytest.mark.parametrize('name', sorted(FIXTURE))
def test_model_config_equals_yaml(name):
    ref, cfg = FIXTURE[name], model_config(name)
    mp = dict(cfg['model_params'])
    assert mp.pop('cam_id_list') == ref['cam_id_list']            # train.py copies dataset_params.cam_id_list into model_params
    assert mp == ref['model_params']
    assert cfg['train_params'] == ref['train_params']
    assert cfg['dataset_params']['cam_id_list'] == ref['cam_id_list']
    assert cfg['dataset_params']['dataset']['name'] == ref['dataset_name']
    assert cfg['dataset_params']['geodesic_param_list'] == ref['geodesic_param_list']


def test_unknown_config_raises():
    with pytest.raises(KeyError):
        model_config('HM36_Nope')
