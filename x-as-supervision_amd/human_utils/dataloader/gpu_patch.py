"""Batch-level GPU replacement of the reference's per-sample CPU loader work
(human_utils/dataloader/dataloader.py:17-91 `generate_patch_sample_data`, :150-191 `generate_item`).

The host keeps what is cheap and scalar (decode is outside this module; the 2x3 crop transform, joint transforms) and
uploads the decoded 8-bit frames once; the GPU does the pixel work for the whole batch: affine crop of image and mask
(OpenCV's fixed-point bilinear arithmetic, bit-exact), MPI mask binarisation, layout / normalisation / background removal,
and the geodesic weight maps.  Output: the batch-dict entries `cam_k_img`, `_mask`, `_geodesic_dis`, `_geodesic_center`,
`_joints`, `_trans_image` the model consumes (dataloader.py:166-191)."""
import ctypes

import numpy as np
import torch

from human_utils.common.imglib.affine import (fliplr_joints, gen_affine_trans_from_box_cv, invert_for_warp, norm_rot_angle,
                                               trans_points_3d)
from human_utils.common.utility.geodesic import compute_geodesic_dis_batch
from xas_amd._lib import call, ptr


def _pack(images, device):
    """list of HxWxC uint8 arrays -> (flat device buffer, offsets [B] int64, sizes [B,2] int32)."""
    offs, hw, total = [], [], 0
    for im in images:
        offs.append(total)
        hw.append(im.shape[:2])
        total += im.size
    flat = torch.empty(total, dtype=torch.uint8)
    for im, o in zip(images, offs):
        flat[o:o + im.size] = torch.from_numpy(np.ascontiguousarray(im).reshape(-1))
    return (flat.to(device, non_blocking=True), torch.tensor(offs, dtype=torch.int64, device=device),
            torch.tensor(hw, dtype=torch.int32, device=device))


def warp_affine_batch(images, trans, patch, device):
    """cv2.warpAffine(img_b, trans_b, (patch, patch), INTER_LINEAR) for every image -> uint8 [B, patch, patch, C]."""
    imgs = [im if im.ndim == 3 else im[..., None] for im in images]
    C = imgs[0].shape[2]
    flat, offs, hw = _pack(imgs, device)
    minv = torch.tensor(np.stack([invert_for_warp(t) for t in trans]), dtype=torch.float64, device=device)
    out = torch.empty(len(imgs), patch, patch, C, dtype=torch.uint8, device=device)
    call('xas_warp_affine_u8', ptr(flat), ptr(offs), ptr(hw), ptr(minv), len(imgs), C, patch, ptr(out))
    return out


def generate_patch_batch(samples, frames, masks, patch_width, patch_height, rect_3d_width, mean, std, device,
                         aug=None, rm_bg=True, mpi_masks=False, geodesic_param_list=(2, 1, 3, 20, 0.0), geodesic_pts=None):
    """samples: list of per-camera database records (dict-like: center_x, center_y, width, height, rot, joints_3d,
    joints_3d_vis, flip_pairs); frames / masks: decoded BGR frames [H,W,3] and masks [H,W] (uint8 numpy);
    aug: optional list of (scale, rot, do_flip, color_scale) per sample (dataloader.py:43-52).
    -> dict with img [B,3,P,P], mask [B,1,P,P], geodesic_dis [B,1,P,P], geodesic_center [B,2], joints [B,K,3],
    trans_image [B,2,3] (device tensors, float32)."""
    if patch_width != patch_height:
        raise NotImplementedError('square patches only (config/*.yaml: 256 x 256)')
    B, P = len(samples), int(patch_width)
    trans, joints_out, imgs, msks, cscale = [], [], [], [], []
    for i, smp in enumerate(samples):
        scale, rot, do_flip, color_scale = aug[i] if aug is not None else (1.0, 0, False, [1.0, 1.0, 1.0])
        rot = norm_rot_angle(rot - smp['rot'] if do_flip else rot + smp['rot'])
        img, msk, c_x = frames[i], masks[i], smp['center_x']
        if do_flip:                                            # affine.py:107-110: only the IMAGE is flipped; the mask is
            img = img[:, ::-1, :]                              # warped un-flipped with the same transform, as the reference
            c_x = img.shape[1] - c_x - 1                       # does (dataloader.py:57-59) - reproduced, not "fixed"
        t = gen_affine_trans_from_box_cv(c_x, smp['center_y'], smp['width'], smp['height'], P, P, scale, rot, False)
        if do_flip:
            j, _ = fliplr_joints(smp['joints_3d'], smp['joints_3d_vis'], img.shape[1], smp['flip_pairs'])
        else:
            j = smp['joints_3d'].copy()
        joints_out.append(trans_points_3d(j, t, 1.0 / (rect_3d_width * scale) * P))
        trans.append(t)
        imgs.append(img)
        msks.append(msk)
        cscale.append(color_scale)
    img_patch = warp_affine_batch(imgs, trans, P, device)                      # [B,P,P,3] BGR uint8
    mask_patch = warp_affine_batch(msks, trans, P, device)[..., 0].contiguous()   # [B,P,P] uint8
    if mpi_masks:                                                             # dataloader.py:62-65
        binm = torch.empty_like(mask_patch)
        call('xas_mask_blur_threshold', ptr(mask_patch), B, P, ptr(binm))
        mask_patch = binm
    out_img = torch.empty(B, 3, P, P, device=device, dtype=torch.float32)
    out_mask = torch.empty(B, 1, P, P, device=device, dtype=torch.float32)
    cs = torch.tensor(cscale, dtype=torch.float32, device=device) if aug is not None else None
    mean3 = (ctypes.c_float * 3)(*[float(v) for v in (mean if mean is not None else (0, 0, 0))])
    std3 = (ctypes.c_float * 3)(*[float(v) for v in (std if std is not None else (1, 1, 1))])
    call('xas_patch_finish', ptr(img_patch), ptr(mask_patch), ptr(cs), ctypes.cast(mean3, ctypes.c_void_p),
         ctypes.cast(std3, ctypes.c_void_p), int(bool(rm_bg)), B, P, ptr(out_img), ptr(out_mask))
    joints = torch.tensor(np.stack(joints_out), dtype=torch.float32, device=device)
    centers = None
    if geodesic_pts is not None and len(geodesic_pts):
        # dataloader.py:189-191: the listed joints (patch pixels) are the sources; geodesic.py:19: astype(int16) truncates
        centers = joints[:, [int(k) for k in geodesic_pts], :2].to(torch.int32).contiguous()
    geo, cen = compute_geodesic_dis_batch(out_mask, geodesic_param_list, centers)
    return {'img': out_img, 'mask': out_mask, 'geodesic_dis': geo, 'geodesic_center': cen, 'joints': joints,
            'trans_image': torch.tensor(np.stack(trans), dtype=torch.float32, device=device)}
