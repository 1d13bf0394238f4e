"""Seeded synthetic inputs for the view-synthesis loss path.

Shapes and value ranges follow the reference's input contract:
images ``uint8/127.5 - 1`` in [-1, 1] (datasets/kitti/kitti_raw_dataset.py:12-14), per-scale
intrinsics (datasets/kitti/kitti_raw_transformed.py:76-93), disparity ``10*sigmoid(.)+0.01``
(models/disp_net.py:7-8,104), pose ``0.01 * mean(conv)`` (models/pose_net.py:52).

Host-side NumPy only; used by bench.py and tests to build identical inputs for the HIP path
and for the CPU oracle.
"""
from __future__ import annotations

import numpy as np

KITTI_FX, KITTI_FY, KITTI_CX, KITTI_CY = 241.7, 246.3, 204.2, 59.0   # 128x416 (data/kitti_raw_loader.py:72-80)


def _resize_align_corners(x, oh, ow):
    """Bilinear, align-corners resize of (B,C,H,W) float32 (what F.resize_images computes,
    models/base_model.py:71-72)."""
    x = np.asarray(x, dtype=np.float32)
    B, C, H, W = x.shape
    u = np.linspace(0, W - 1, num=ow).astype(np.float32)
    v = np.linspace(0, H - 1, num=oh).astype(np.float32)
    u0 = np.clip(np.floor(u).astype(np.int32), 0, max(W - 2, 0))
    v0 = np.clip(np.floor(v).astype(np.int32), 0, max(H - 2, 0))
    u1 = np.minimum(u0 + 1, W - 1)
    v1 = np.minimum(v0 + 1, H - 1)
    wu1 = (u - u0).astype(np.float32)
    wv1 = (v - v0).astype(np.float32)
    wu0 = np.float32(1) - wu1
    wv0 = np.float32(1) - wv1
    r0 = x[:, :, v0]
    r1 = x[:, :, v1]
    top = r0[:, :, :, u0] * wu0 + r0[:, :, :, u1] * wu1
    bot = r1[:, :, :, u0] * wu0 + r1[:, :, :, u1] * wu1
    return (top * wv0[:, None] + bot * wv1[:, None]).astype(np.float32)


def image_pyramid(x, n_scales):
    """[(B,C,H>>s,W>>s)] for s in 0..n_scales-1 (base_model.py:70-72)."""
    B, C, H, W = x.shape
    return [np.ascontiguousarray(_resize_align_corners(x, H // (2 ** s), W // (2 ** s)))
            if s else np.ascontiguousarray(x, dtype=np.float32) for s in range(n_scales)]


def multi_scale_intrinsics(K, n_scales):
    """(B,3,3) -> (B,S,3,3): fx, fy, cx, cy divided by 2**s (kitti_raw_transformed.py:86-91)."""
    K = np.asarray(K, dtype=np.float32)
    out = np.zeros((K.shape[0], n_scales, 3, 3), dtype=np.float32)
    for s in range(n_scales):
        f = np.float32(2 ** s)
        out[:, s, 0, 0] = K[:, 0, 0] / f
        out[:, s, 1, 1] = K[:, 1, 1] / f
        out[:, s, 0, 2] = K[:, 0, 2] / f
        out[:, s, 1, 2] = K[:, 1, 2] / f
        out[:, s, 2, 2] = 1
    return out


def _smooth_field(rng, B, C, H, W, div):
    lh, lw = max(H // div, 2), max(W // div, 2)
    low = rng.uniform(-1.0, 1.0, size=(B, C, lh, lw)).astype(np.float32)
    return _resize_align_corners(low, H, W)


def _shift_replicate(x, sx, sy):
    """x shifted by (sx, sy) pixels like np.roll, but with the border pixels REPLICATED instead of wrapped around: no seam."""
    H, W = x.shape[-2:]
    xi = np.clip(np.arange(W) - sx, 0, W - 1)
    yi = np.clip(np.arange(H) - sy, 0, H - 1)
    return x[..., yi, :][..., xi]


def make_inputs(B=2, H=128, W=416, n_src=2, n_scales=4, seed=1, with_masks=False,
                rot_sigma=0.01, trans_sigma=0.02, seam="roll", disp_div=4, disp_noise=0.1):
    """Returns a dict of float32 C-contiguous arrays:

    tgt (B,3,H,W); src (B,n_src,3,H,W); tgt_pyr[s] (B,3,h,w); src_pyr[s] (B,3n,h,w);
    intrinsics (B,S,3,3); disps[s] (B,1,h,w); poses[i] (B,6); masks[s] (B,n,h,w) | None

    seam: how a source is made from the target pattern "shifted by a few px" (SURVEY.md 8(d)): "roll" wraps the pattern around
    (np.roll: the sources carry a seam of full contrast a few pixels from two of their borders -- the inputs of rounds 1-4,
    kept: a step edge is the hardest case for a bilinear sampler's position accuracy); "shift" replicates the border instead
    (no seam: image-like texture everywhere).  Same random draws in both, so everything else is identical.

    disp_div, disp_noise: the disparity logit is uniform noise at 1/disp_div of the scale's resolution, bilinearly upsampled, plus
    disp_noise * N(0,1) per pixel.  The defaults (4, 0.1: the inputs of every round's headline) make a ROUGH field -- the disparity
    swings over most of its range within four pixels, so that neighbouring samples' taps lie several source rows apart (what a
    bilinear gather and the optional d_src scatter pay for: profiles/r05_d_src.txt).  (32, 0.0) is a field as smooth as a network's
    disparity map away from object boundaries; bench.py reports both (`*_smooth_disp`).  Same random draws either way.
    """
    if seam not in ("roll", "shift"):
        raise ValueError("seam must be 'roll' or 'shift', got %r" % (seam,))
    rng = np.random.RandomState(seed)
    tex = 1.4 * _smooth_field(rng, B, 3, H, W, 8) + 0.2 * _smooth_field(rng, B, 3, H, W, 2)
    tgt = np.clip(tex, -1, 1).astype(np.float32)
    srcs = []
    for i in range(n_src):
        sx = int(rng.randint(-4, 5))
        sy = int(rng.randint(-2, 3))
        shifted = np.roll(np.roll(tex, sx, axis=3), sy, axis=2) if seam == "roll" else _shift_replicate(tex, sx, sy)
        noise = 0.05 * rng.standard_normal(size=tex.shape).astype(np.float32)
        srcs.append(np.clip(shifted + noise, -1, 1).astype(np.float32))
    src = np.stack(srcs, axis=1)
    # keep every value away from exactly 0 (the reference's mask is `== 0`, base_model.py:96)
    tgt[tgt == 0] = np.float32(1e-3)
    src[src == 0] = np.float32(1e-3)
    stacked = src.reshape(B, 3 * n_src, H, W)
    tgt_pyr = image_pyramid(tgt, n_scales)
    src_pyr = image_pyramid(stacked, n_scales)

    scale = W / 416.0
    K = np.zeros((B, 3, 3), dtype=np.float32)
    jit = 1.0 + 0.05 * rng.uniform(-1, 1, size=(B, 4)).astype(np.float32)
    K[:, 0, 0] = KITTI_FX * scale * jit[:, 0]
    K[:, 1, 1] = KITTI_FY * scale * jit[:, 1]
    K[:, 0, 2] = KITTI_CX * scale * jit[:, 2]
    K[:, 1, 2] = KITTI_CY * scale * jit[:, 3]
    K[:, 2, 2] = 1
    intrinsics = multi_scale_intrinsics(K, n_scales)

    disps, masks = [], []
    for s in range(n_scales):
        h, w = H // (2 ** s), W // (2 ** s)
        n = _smooth_field(rng, B, 1, h, w, disp_div) * 1.5 + np.float32(disp_noise) * rng.standard_normal(size=(B, 1, h, w)).astype(np.float32)
        disps.append(np.ascontiguousarray(10.0 / (1.0 + np.exp(-n)) + 0.01, dtype=np.float32))
        if with_masks:
            masks.append(np.ascontiguousarray(
                _smooth_field(rng, B, n_src, h, w, 4) * 2 + 0.1 * rng.standard_normal(size=(B, n_src, h, w)),
                dtype=np.float32))
    poses = []
    for i in range(n_src):
        p = np.concatenate([rot_sigma * rng.standard_normal(size=(B, 3)),
                            trans_sigma * rng.standard_normal(size=(B, 3))], axis=1).astype(np.float32)
        poses.append(np.ascontiguousarray(p))
    return dict(tgt=tgt, src=src, tgt_pyr=tgt_pyr, src_pyr=src_pyr, intrinsics=intrinsics,
                disps=disps, poses=poses, masks=masks if with_masks else None,
                B=B, H=H, W=W, n_src=n_src, n_scales=n_scales)
