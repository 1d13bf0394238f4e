"""On-device data augmentation of a training batch: the reference's `data_augmentation`
(datasets/kitti/kitti_raw_transformed.py:23-74) -- random scaling, random crop, random horizontal
flip and the matching intrinsics update -- followed by `get_multi_scale_intrinsics` (:76-93).

The random draws stay on the host, in the reference's order per sample
(`np.random.uniform(1, 1.15, 2)`, two `np.random.randint`, one `np.random.rand`), so a seeded run sees
the same parameters; the image work (resize + crop + flip of every frame) is one gather kernel
(`sfm_augment_fwd`) instead of three array passes per sample in the data-loader processes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import ops
from ._lib import check, lib

__all__ = ["sample_params", "augment_intrinsics", "augment_images", "data_augmentation", "get_multi_scale_intrinsics"]


def sample_params(rng, B, H, W):
    """(B,7) float64: x_scaling, y_scaling, scaled_h, scaled_w, offset_y, offset_x, flip -- drawn per
    sample in the reference's call order (kitti_raw_transformed.py:34, :50-51, :64)."""
    out = np.zeros((B, 7), dtype=np.float64)
    for b in range(B):
        scaling = rng.uniform(1, 1.15, 2)                                  # :34
        x_scaling, y_scaling = scaling[0], scaling[1]
        sh, sw = int(H * y_scaling), int(W * x_scaling)                    # :37-38
        oy = int(rng.randint(0, sh - H + 1))                               # :50
        ox = int(rng.randint(0, sw - W + 1))                               # :51
        flip = 1.0 if rng.rand() < 0.5 else 0.0                            # :64
        out[b] = (x_scaling, y_scaling, sh, sw, oy, ox, flip)
    return out


def augment_intrinsics(K, params, W):
    """(B,3,3) float32 intrinsics after scaling (:41-44), cropping (:54-57) and flipping (:66)."""
    K = np.asarray(K, dtype=np.float32)
    out = np.zeros_like(K)
    for b in range(K.shape[0]):
        xs, ys, _, _, oy, ox, flip = params[b]
        fx = K[b, 0, 0] * xs
        fy = K[b, 1, 1] * ys
        cx = K[b, 0, 2] * xs
        cy = K[b, 1, 2] * ys
        m = np.array([[fx, 0., cx], [0., fy, cy], [0., 0., 1.]], dtype='f')       # make_intrinsics_matrix :16-21
        m = np.array([[m[0, 0], 0., m[0, 2] - int(ox)], [0., m[1, 1], m[1, 2] - int(oy)], [0., 0., 1.]], dtype='f')
        if flip:
            m[0, 2] = W - m[0, 2]
        out[b] = m
    return out


def get_multi_scale_intrinsics(K, n_scales):
    """kitti_raw_transformed.py:76-93, batched: (B,3,3) -> (B,S,3,3)"""
    K = np.asarray(K, dtype=np.float32)
    out = np.zeros((K.shape[0], n_scales, 3, 3), dtype=np.float32)
    for s in range(n_scales):
        out[:, s, 0, 0] = K[:, 0, 0] / (2 ** s)
        out[:, s, 1, 1] = K[:, 1, 1] / (2 ** s)
        out[:, s, 0, 2] = K[:, 0, 2] / (2 ** s)
        out[:, s, 1, 2] = K[:, 1, 2] / (2 ** s)
        out[:, s, 2, 2] = 1
    return out


def augment_images(imgs, params):
    """imgs: (B,F,3,H,W) device array (target + sources); params from `sample_params` -> same shape."""
    imgs = ops._dev(imgs, "imgs", 5)
    B, F, Cc, H, W = imgs.shape
    p = torch.from_numpy(np.ascontiguousarray(np.asarray(params)[:, 2:7], dtype=np.float32)).to(imgs.device)
    out = torch.empty_like(imgs)
    with torch.cuda.device(imgs.device):
        check(lib.sfm_augment_fwd(C.c_void_p(imgs.data_ptr()), C.c_void_p(p.data_ptr()), C.c_void_p(out.data_ptr()),
                                  B, F, Cc, H, W, ops._stream()))
    return out


def data_augmentation(tgt_img, src_imgs, intrinsics, rng=np.random, n_scales=4):
    """Batched `_transform` (kitti_raw_transformed.py:95-102): tgt (B,3,H,W), src (B,S,3,H,W), K (B,3,3)
    -> (tgt, src, multi-scale intrinsics (B,n_scales,3,3) as a host array)."""
    B, _, H, W = tgt_img.shape
    params = sample_params(rng, B, H, W)
    imgs = torch.cat([tgt_img[:, None], src_imgs], dim=1)                   # :70
    aug = augment_images(imgs, params)
    K = augment_intrinsics(intrinsics, params, W)
    return aug[:, 0], aug[:, 1:], get_multi_scale_intrinsics(K, n_scales)
