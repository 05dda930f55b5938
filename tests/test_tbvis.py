"""Logging contract of train_util.tb_vis: the product's function must issue exactly the (kind, tag, step) calls the
REFERENCE's tb_vis issues for the same batch / output dictionaries (tests/golden/tbvis_tags.json, written by executing
the reference function, train_util.py:229-305, with recording stand-ins for the writer and image helpers)."""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class Rec:
    def __init__(self):
        self.calls, self.images = [], {}

    def add_scalar(self, tag, value, step):
        float(np.asarray(value).reshape(-1)[0])
        self.calls.append(['scalar', tag, int(step)])

    def add_image(self, tag, img, step):
        self.calls.append(['image', tag, int(step)])
        self.images[tag] = np.asarray(img)

    def add_text(self, tag, text, step):
        self.calls.append(['text', tag, int(step)])


class Sched:
    def get_last_lr(self):
        return [2e-4]


def _inputs():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_mg_inputs', os.path.join(HERE, 'golden', 'make_golden_inputs.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.tbvis_inputs()


def test_tb_vis_issues_the_reference_calls():
    from train_util import tb_vis
    with open(os.path.join(HERE, 'golden', 'tbvis_tags.json')) as f:
        gold = json.load(f)
    torch.manual_seed(0)
    x, out, losses = _inputs()
    cfg = {'dataset_params': {'dataiter': {'mean': [0.0, 0.0, 0.0], 'std': [255.0, 255.0, 255.0]}}}
    for step in (50, 51):
        w = Rec()
        tb_vis(w, step, np.array([[1, 4]]), np.arange(18), 1.25, losses, torch.tensor(0.5), out, x, cfg, Sched())
        assert w.calls == gold[str(step)], step
        if step == 50:
            for tag, im in w.images.items():
                assert im.dtype == np.uint8 and im.ndim == 3 and im.shape[0] == 3 or im.shape[0] == 1, tag
            ov = w.images['training_pose_2d/cam_0_gt_pose']
            assert ov.shape == (3, 32, 32) and ov.std() > 0
    w = Rec()
    tb_vis(w, 100, np.array([[1, 4]]), np.arange(18), None, {}, None, out, x, cfg, Sched())
    assert w.calls == gold['100_no_gen']


def test_jsonl_writer(tmp_path):
    from train_util import JsonlWriter
    w = JsonlWriter(str(tmp_path))
    w.add_scalar('a/b', torch.tensor(1.5), 3)
    w.add_text('t', 'hello', 3)
    w.add_image('i/j', np.zeros((3, 4, 4), np.uint8), 3)
    w.close()
    lines = [json.loads(l) for l in open(os.path.join(str(tmp_path), 'scalars.jsonl'))]
    assert lines[0] == {'tag': 'a/b', 'step': 3, 'value': 1.5} and lines[1]['text'] == 'hello'
    assert os.path.exists(os.path.join(str(tmp_path), 'images', 'i__j_00000003.npy'))
