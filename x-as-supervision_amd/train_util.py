"""Training-time logging (reference: train_util.py:116-305): `tb_vis` with the reference's signature, tags and cadence
(scalars every step, text / images every 50 steps), and its image helpers.  The reference draws with OpenCV and
matplotlib; OpenCV is not a dependency here: skeleton overlays are rasterised with numpy, the 3-D skeleton / weight-map /
depth panels use matplotlib when it is importable and are skipped otherwise.  `tb_log` is any object with the
SummaryWriter methods add_scalar / add_image / add_text (torch.utils.tensorboard.SummaryWriter, or JsonlWriter below
when TensorBoard is not installed).  Only rank 0 logs (train.py:196-199)."""
import json
import os

import numpy as np
import torch


class JsonlWriter:
    """Minimal stand-in for SummaryWriter: scalars and text go to <log_dir>/scalars.jsonl, images to PNG-less .npy files
    under <log_dir>/images (uint8 CHW), one per (tag, step)."""

    def __init__(self, log_dir):
        self.log_dir = log_dir
        os.makedirs(os.path.join(log_dir, 'images'), exist_ok=True)
        self._f = open(os.path.join(log_dir, 'scalars.jsonl'), 'a')

    def add_scalar(self, tag, value, step):
        self._f.write(json.dumps({'tag': tag, 'step': int(step), 'value': float(np.asarray(value).reshape(-1)[0])}) + '\n')

    def add_text(self, tag, text, step):
        self._f.write(json.dumps({'tag': tag, 'step': int(step), 'text': str(text)}) + '\n')

    def add_image(self, tag, img, step):
        np.save(os.path.join(self.log_dir, 'images', '%s_%08d.npy' % (tag.replace('/', '__'), int(step))), np.asarray(img))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def img_vis(img, mean=None, std=None):
    """CHW float image -> uint8 (train_util.py:154-165)."""
    img = _np(img).copy()
    if mean and std:
        for i in range(len(img)):
            img[i, :, :] = img[i, :, :] * std[i] + mean[i]
    if np.max(img) < 128:
        img = img * 255
    return np.uint8(img)


def _draw_line(img, p0, p1, color):
    n = int(max(abs(p1[0] - p0[0]), abs(p1[1] - p0[1]))) + 1
    xs = np.rint(np.linspace(p0[0], p1[0], n)).astype(int)
    ys = np.rint(np.linspace(p0[1], p1[1], n)).astype(int)
    ok = (xs >= 0) & (xs < img.shape[1]) & (ys >= 0) & (ys < img.shape[0])
    img[ys[ok], xs[ok]] = color


def _draw_dot(img, p, color, r=2):
    x, y = int(round(p[0])), int(round(p[1]))
    img[max(0, y - r):min(img.shape[0], y + r + 1), max(0, x - r):min(img.shape[1], x + r + 1)] = color


def pose_vis(pose, size, flip_pairs, parent_ids=None, is_gt=False, img=None, mean=None, std=None):
    """2-D skeleton overlay, CHW uint8 (train_util.py:116-138; numpy rasteriser instead of cv2 drawing)."""
    pose = np.array(pose, dtype=np.float64)
    if not is_gt:
        pose = (pose + 1) / 2.0
        pose[:, 0] *= (size[0] - 1)
        pose[:, 1] *= (size[1] - 1)
    if img is None:
        canvas = np.ones([size[0], size[1], 3]) * 255.0
    else:
        canvas = _np(img).copy()
        if mean is not None and std is not None:
            for i in range(len(canvas)):
                canvas[i, :, :] = canvas[i, :, :] * std[i] + mean[i]
        canvas = canvas.transpose([1, 2, 0])
        if np.max(canvas) < 128:
            canvas = canvas * 255.0
    canvas = np.ascontiguousarray(canvas, dtype=np.uint8)
    left = {p[0] for p in flip_pairs} if flip_pairs is not None and np.max(flip_pairs) < pose.shape[0] else set()
    if parent_ids is not None:
        for j, p in enumerate(parent_ids):
            if 0 <= p < pose.shape[0] and p != j:
                _draw_line(canvas, pose[j, :2], pose[p, :2], (0, 160, 0))
    for j in range(pose.shape[0]):
        _draw_dot(canvas, pose[j, :2], (255, 0, 0) if j in left else (0, 0, 255))
    return canvas.transpose([2, 0, 1])


def _figure_to_chw(fig):
    fig.canvas.draw()
    buf = np.asarray(fig.canvas.buffer_rgba())[..., :3]
    import matplotlib.pyplot as plt
    plt.close(fig)
    return np.ascontiguousarray(buf).transpose([2, 0, 1])


def _plt():
    try:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        return plt
    except Exception:
        return None


def pose_vis_3d(keypoints_3d, flip_pairs, parent_ids=None, ref_keypoints=None):
    """3-D skeleton panel (train_util.py:140-152)."""
    plt = _plt()
    if plt is None:
        return None
    fig = plt.figure()
    ax = fig.add_subplot(111, projection='3d')
    for kp, col in ((keypoints_3d, 'b'), (ref_keypoints, 'k')):
        if kp is None:
            continue
        kp = np.asarray(kp)
        if parent_ids is not None:
            for j, p in enumerate(parent_ids):
                if 0 <= p < kp.shape[0] and p != j:
                    ax.plot([kp[j, 0], kp[p, 0]], [kp[j, 2], kp[p, 2]], [-kp[j, 1], -kp[p, 1]], c=col)
        ax.scatter(kp[:, 0], kp[:, 2], -kp[:, 1], c=col, s=6)
    return _figure_to_chw(fig)


def dis_vis(distance, centers):
    """Geodesic weight map with its source pixels (train_util.py:167-183)."""
    plt = _plt()
    if plt is None:
        return None
    fig = plt.figure()
    plt.imshow(distance[0], interpolation='nearest')
    for c in np.asarray(centers).reshape(-1, 2):
        plt.scatter(c[0], c[1], c='r', s=5)
    return _figure_to_chw(fig)


def depth_heatmap_vis(depth_map, gt_pose_2d, depth_scale=256, heat_w=6, heat_h=6):
    """Per-joint depth distributions with the ground-truth bin marked (train_util.py:185-227)."""
    plt = _plt()
    if plt is None:
        return None
    import matplotlib.colors as mcolors
    K, H = depth_map.shape
    gt_depth = np.clip(((gt_pose_2d[:, [2]] / depth_scale) + 1) / 2, 0, 1) * H
    cmap = mcolors.ListedColormap(['white', 'red'])
    norm = mcolors.BoundaryNorm([0, 1, 2], cmap.N)
    fig, axes = plt.subplots(nrows=heat_h, ncols=heat_w, figsize=(10, 4))
    for i in range(K):
        line = np.tile(depth_map[[i]], (10, 1))
        mask = np.zeros_like(line)
        loc = min(H - 1, int(gt_depth[i, 0]))
        mask[:, loc] = 1.0
        mask[:, max(0, loc - 1)] = 1.0
        a, b = axes[i // heat_w * 2, i % heat_w], axes[(i // heat_w) * 2 + 1, i % heat_w]
        a.imshow(line, cmap='Reds')
        b.imshow(mask, cmap=cmap, norm=norm)
        for ax in (a, b):
            ax.set_xticks([])
            ax.set_yticks([])
    plt.tight_layout()
    return _figure_to_chw(fig)


def _image(tb_log, tag, img, step):
    if img is not None:
        tb_log.add_image(tag, img, step)


def tb_vis(tb_log, cur_step, tb_pair_ids, tb_parent_ids, total_loss, loss_kp, loss_disc, output, x, config,
           scheduler_detector, simple_version=False):
    """Same tags, same cadence as the reference (train_util.py:229-305)."""
    if not simple_version:
        if total_loss is not None:
            tb_log.add_scalar('training_loss/total_loss', total_loss, cur_step)
        for key, value in loss_kp.items():
            tb_log.add_scalar('training_loss/{}'.format(key), _np(value.mean()), cur_step)
        tb_log.add_scalar('meta/learning_rate/detector', float(scheduler_detector.get_last_lr()[0]), cur_step)
        for key in output.keys():
            if key.startswith('line_width'):
                for i, val in enumerate(output[key]):
                    tb_log.add_scalar('training_line_width/{}_{}'.format(key, i), _np(val), cur_step)
        if loss_disc is not None:
            tb_log.add_scalar('training_loss/smpl_disc', _np(loss_disc), cur_step)
    if cur_step % 50 != 0:
        return
    di = config['dataset_params'].get('dataiter') if 'dataset_params' in config else None
    mean, std = (di['mean'], di['std']) if di else (None, None)
    if not simple_version:
        tb_log.add_text('training_img/file_name', '{}'.format(x['cam_0_img_path'][0]), cur_step)
    else:
        tb_log.add_text('training_img/file_name_2d', '{}'.format(x['cam_mono_img_path'][0]), cur_step)
    for key in x.keys():
        if 'pseudo' in key:
            continue
        if key.endswith('img'):
            tb_log.add_image('training_img/{}'.format(key), img_vis(x[key][0], mean=mean, std=std), cur_step)
        elif key.endswith('mask'):
            tb_log.add_image('training_mask/{}'.format(key), img_vis(x[key][0]), cur_step)
        elif key.endswith('joints'):
            cam_key = key.split('_joints')[0]
            tb_log.add_image('training_pose_2d/{}_gt_pose'.format(cam_key),
                             pose_vis(_np(x[key][0, :, :2]), x['{}_img'.format(cam_key)].shape[2:4], tb_pair_ids, tb_parent_ids,
                                      img=x['{}_img'.format(cam_key)][0].clone(), mean=mean, std=std, is_gt=True), cur_step)
        elif key.endswith('geodesic_dis'):
            cam_key = key.split('_geodesic_dis')[0]
            _image(tb_log, 'training_weight/{}'.format(key),
                   dis_vis(_np(x[key][0]), _np(x['{}_geodesic_center'.format(cam_key)][0])), cur_step)
    for key in output.keys():
        if key.startswith('mask'):
            tb_log.add_image('training_mask/{}'.format(key), img_vis(output[key][0]), cur_step)
        elif key.startswith('pose_2d'):
            mode = key.split('pose_2d_pred_')[1].split('_ori')[0]
            page = 'training_pose_2d' if 'pseudo' not in key else 'training_pseudo'
            tb_log.add_image('{}/{}'.format(page, key),
                             pose_vis(_np(output[key][0, :, :2]), x['{}_img'.format(mode)].shape[2:4], tb_pair_ids, tb_parent_ids,
                                      img=x['{}_img'.format(mode)][0].clone(), mean=mean, std=std), cur_step)
        elif key.startswith('pose_3d'):
            page = 'training_pose_3d' if 'pseudo' not in key else 'training_pseudo'
            _image(tb_log, '{}/{}'.format(page, key), pose_vis_3d(_np(output[key][0]), tb_pair_ids, tb_parent_ids), cur_step)
        elif key.startswith('pose_smpl_2d') and not simple_version:
            tb_log.add_image('training_smpl/{}'.format(key),
                             pose_vis(_np(output[key][0, :, :2]), x['cam_0_img'].shape[2:4], tb_pair_ids, tb_parent_ids), cur_step)
        elif key.startswith('pose_smpl_3d') and not simple_version:
            _image(tb_log, 'training_smpl/{}'.format(key), pose_vis_3d(_np(output[key][0]), tb_pair_ids, tb_parent_ids), cur_step)
        elif key.startswith('depth_map') and not simple_version:
            mode = key.split('depth_map_')[1]
            _image(tb_log, 'training_depth/{}'.format(key),
                   depth_heatmap_vis(_np(output[key]), _np(x['{}_joints'.format(mode)][0])), cur_step)
        elif 'logits' in key and not simple_version:
            tb_log.add_scalar('training_disc/{}'.format(key), _np(output[key][0, ...]), cur_step)
    if 'kp_gt_world' in output.keys():
        _image(tb_log, 'training_pose_3d/src_gt_pose_3d', pose_vis_3d(_np(output['kp_gt_world'][0]), tb_pair_ids, tb_parent_ids),
               cur_step)


# names this mirror does not replace resolve, lazily, to the reference module behind it on sys.path
from xas_amd._next import fallthrough as _fallthrough  # noqa: E402

__getattr__ = _fallthrough(__name__, __file__)
