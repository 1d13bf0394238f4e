"""CPU oracle (NumPy restatement) of SfM-Learner's photometric view-synthesis loss path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sfm-learner-chainer_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and there only as the checker / the CPU baseline.

What it restates (citations are into /root/reference):

* geometry            models/transform.py:11-193
* loss loop           models/base_model.py:57-124
* SSIM / smoothness   models/base_model.py:126-185
* alternative sampler models/spational_transformer_sampler_interp.py:32-149
* input contract      datasets/kitti/kitti_raw_transformed.py:76-93,
                      datasets/kitti/kitti_raw_dataset.py:12-14

The arithmetic of the live path sits in a third-party dependency that is NOT under
/root/reference: ``chainer==4.0.0b1`` (requirements.txt:1).  Its ops
(``F.spatial_transformer_sampler``, ``F.resize_images``, ``F.average_pooling_2d``,
``F.batch_matmul``, ``F.batch_inv`` ...) are restated here from their published
definitions.

Pinning status
--------------
* ``interp_sampler_forward/backward`` (A8') is PINNED: tests/golden/interp_sampler_*.npz
  were produced by executing the reference's own file (see tests/golden/make_golden.py).
* ``euler2mat`` is PINNED against kitti_eval/odom_util.py:167-200 (same X.Y.Z product),
  tests/golden/euler_odom_util.npz.
* Everything that goes through Chainer's own ops is "parity unpinned": the reference ships
  no tests, golden vectors or fixtures for this path and Chainer is not installable here.
  Those parts are pinned only by analytic known-answer cases and by
  fp64 finite-difference checks of the hand-derived backward.

All functions take an explicit ``dtype`` (np.float32 = the reference's precision,
np.float64 = for finite-difference gradient checks).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "euler2mat", "pose_vec2mat", "proj_tgt_to_src", "batch_inv3", "generate_2dmeshgrid",
    "pixel2cam", "cam2pixel", "spatial_transformer_sampler", "spatial_transformer_sampler_backward",
    "interp_sampler_forward", "interp_sampler_backward", "projective_inverse_warp",
    "projective_inverse_warp_backward", "resize_images", "average_pooling_3x3",
    "compute_ssim", "compute_smooth_loss", "compute_disp_smooth", "compute_exp_reg_loss",
    "get_multi_scale_intrinsics", "sfm_loss", "SfmLossResult", "data_augmentation",
]


# --------------------------------------------------------------------------------------
# small batched linear algebra, written out so that the operation order is explicit
# (Chainer's F.batch_matmul defers to BLAS, whose summation order is unspecified)
# --------------------------------------------------------------------------------------
def _bmm(a, b):
    """Per-sample matmul (F.batch_matmul).  a: (N,m,k)  b: (N,k,n) -> (N,m,n).
    Left-to-right accumulation over k, no fused multiply-add."""
    N, m, k = a.shape
    n = b.shape[2]
    out = a[:, :, 0:1] * b[:, 0:1, :]
    for j in range(1, k):
        out = out + a[:, :, j:j + 1] * b[:, j:j + 1, :]
    return out


def batch_inv3(K):
    """F.batch_inv for (N,3,3) (transform.py:105).  Adjugate / determinant."""
    a, b, c = K[:, 0, 0], K[:, 0, 1], K[:, 0, 2]
    d, e, f = K[:, 1, 0], K[:, 1, 1], K[:, 1, 2]
    g, h, i = K[:, 2, 0], K[:, 2, 1], K[:, 2, 2]
    A = e * i - f * h
    B = -(d * i - f * g)
    C = d * h - e * g
    det = a * A + b * B + c * C
    inv = np.empty_like(K)
    inv[:, 0, 0] = A
    inv[:, 0, 1] = -(b * i - c * h)
    inv[:, 0, 2] = b * f - c * e
    inv[:, 1, 0] = B
    inv[:, 1, 1] = a * i - c * g
    inv[:, 1, 2] = -(a * f - c * d)
    inv[:, 2, 0] = C
    inv[:, 2, 1] = -(a * h - b * g)
    inv[:, 2, 2] = a * e - b * d
    return inv / det[:, None, None]


# --------------------------------------------------------------------------------------
# pose -> projection   (transform.py:11-91)
# --------------------------------------------------------------------------------------
def _euler_parts(r, dtype):
    """transform.py:21-37.  r: (N,3) -> clipped r, cos, sin, zmat, ymat, xmat."""
    r = np.asarray(r, dtype=dtype)
    N = r.shape[0]
    pi = dtype(np.pi)
    rc = np.clip(r, -pi, pi)                                    # :23
    cr, sr = np.cos(rc), np.sin(rc)                             # :24-25
    zeros = np.zeros(N, dtype=dtype)
    ones = np.ones(N, dtype=dtype)
    zmat = np.stack([cr[:, 2], -sr[:, 2], zeros,
                     sr[:, 2], cr[:, 2], zeros,
                     zeros, zeros, ones], axis=1).reshape(N, 3, 3)   # :27-29
    ymat = np.stack([cr[:, 1], zeros, sr[:, 1],
                     zeros, ones, zeros,
                     -sr[:, 1], zeros, cr[:, 1]], axis=1).reshape(N, 3, 3)  # :31-33
    xmat = np.stack([ones, zeros, zeros,
                     zeros, cr[:, 0], -sr[:, 0],
                     zeros, sr[:, 0], cr[:, 0]], axis=1).reshape(N, 3, 3)   # :35-37
    return rc, cr, sr, zmat, ymat, xmat


def euler2mat(r, dtype=np.float32):
    """transform.py:11-40.  R = (X . Y) . Z"""
    _, _, _, zmat, ymat, xmat = _euler_parts(r, dtype)
    return _bmm(_bmm(xmat, ymat), zmat)                         # :39


def pose_vec2mat(vec, dtype=np.float32):
    """transform.py:43-59.  vec: (N,6) rx,ry,rz,tx,ty,tz -> (N,4,4)"""
    vec = np.asarray(vec, dtype=dtype)
    N = vec.shape[0]
    R = euler2mat(vec[:, :3], dtype)
    T = np.zeros((N, 4, 4), dtype=dtype)
    T[:, :3, :3] = R
    T[:, :3, 3] = vec[:, 3:]
    T[:, 3, 3] = 1
    return T


def _K4(K, dtype):
    K = np.asarray(K, dtype=dtype)
    N = K.shape[0]
    K4 = np.zeros((N, 4, 4), dtype=dtype)                        # transform.py:86
    K4[:, :3, :3] = K
    K4[:, 3, 3] = 1
    return K4


def proj_tgt_to_src(vec, K, dtype=np.float32):
    """transform.py:64-91 without the host/device hop.  -> (N,4,4)  Pm = K4 . T"""
    return _bmm(_K4(K, dtype), pose_vec2mat(vec, dtype))        # :88


def proj_tgt_to_src_backward(vec, K, gPm, dtype=np.float32):
    """Hand-derived backward of proj_tgt_to_src.  gPm: (N,4,4) -> (N,6)"""
    vec = np.asarray(vec, dtype=dtype)
    gPm = np.asarray(gPm, dtype=dtype)
    K4 = _K4(K, dtype)
    rc, cr, sr, zmat, ymat, xmat = _euler_parts(vec[:, :3], dtype)
    gT = _bmm(np.transpose(K4, (0, 2, 1)), gPm)
    gt = gT[:, :3, 3]
    gR = gT[:, :3, :3]
    xy = _bmm(xmat, ymat)
    gXY = _bmm(gR, np.transpose(zmat, (0, 2, 1)))
    gZ = _bmm(np.transpose(xy, (0, 2, 1)), gR)
    gX = _bmm(gXY, np.transpose(ymat, (0, 2, 1)))
    gY = _bmm(np.transpose(xmat, (0, 2, 1)), gXY)
    g_cos = np.stack([gX[:, 1, 1] + gX[:, 2, 2],
                      gY[:, 0, 0] + gY[:, 2, 2],
                      gZ[:, 0, 0] + gZ[:, 1, 1]], axis=1)
    g_sin = np.stack([gX[:, 2, 1] - gX[:, 1, 2],
                      gY[:, 0, 2] - gY[:, 2, 0],
                      gZ[:, 1, 0] - gZ[:, 0, 1]], axis=1)
    g_rc = -sr * g_cos + cr * g_sin
    pi = dtype(np.pi)
    inside = (vec[:, :3] > -pi) & (vec[:, :3] < pi)             # F.clip backward
    g_r = np.where(inside, g_rc, dtype(0))
    return np.concatenate([g_r, gt], axis=1).astype(dtype)


# --------------------------------------------------------------------------------------
# pixel <-> camera   (transform.py:94-154)
# --------------------------------------------------------------------------------------
def generate_2dmeshgrid(H, W, N, dtype=np.float32):
    """transform.py:137-154.  (N,3,H*W) rows: x (fastest), y, 1."""
    ys, xs = np.meshgrid(np.arange(0, H, dtype=dtype), np.arange(0, W, dtype=dtype), indexing="ij")
    grid = np.concatenate([xs[None], ys[None], np.ones((1, H, W), dtype=dtype)], axis=0)
    return np.broadcast_to(grid.reshape(1, 3, H * W), (N, 3, H * W))


def pixel2cam(depthes, pixel_coords, intrinsics, dtype=np.float32):
    """transform.py:94-109.  depthes (N,3,P) * (K^-1 . pix) ; append ones -> (N,4,P)"""
    N, _, P = depthes.shape
    ray = _bmm(batch_inv3(np.asarray(intrinsics, dtype=dtype)), np.asarray(pixel_coords, dtype=dtype))  # :105
    cam = depthes * ray                                                                                  # :107
    return np.concatenate([cam, np.ones((N, 1, P), dtype=dtype)], axis=1), ray                           # :108


def cam2pixel(cam_coords, proj, H, W, dtype=np.float32):
    """transform.py:111-133.  Returns (p_s_xy (N,2,H,W), aux dict for backward/tests)."""
    N = cam_coords.shape[0]
    q = _bmm(proj, cam_coords)                                   # :122
    z = q[:, 2:3, :] + dtype(1e-10)                              # :123
    half_w = dtype((W - 1) / 2.)
    half_h = dtype((H - 1) / 2.)
    U = q[:, 0:1] / z
    V = q[:, 1:2] / z
    xn = U / half_w - dtype(1)                                   # :124
    yn = V / half_h - dtype(1)                                   # :125
    p = np.concatenate([xn, yn], axis=1)                         # :126
    inside = (p > -1) & (p < 1)                                  # :129
    mask = np.where(inside, dtype(1), dtype(2))                  # :128,130
    p2 = p * mask                                                # :131
    # distance of the un-doubled coordinate to the decision boundary |.|=1, for knife-edge
    # bookkeeping in parity tests (not part of the reference)
    margin = np.abs(np.abs(p) - 1).min(axis=1).reshape(N, H, W)
    aux = dict(q=q, z=z, U=U, V=V, mask=mask, margin=margin)
    return p2.reshape(N, 2, H, W), aux                           # :132


# --------------------------------------------------------------------------------------
# Chainer's F.spatial_transformer_sampler (call site transform.py:189) -- restated, [recalled]
# --------------------------------------------------------------------------------------
def _sampler_coords(grid, H, W, dtype):
    B = grid.shape[0]
    g = grid.reshape(B, 2, -1)
    u = g[:, 0]
    v = g[:, 1]
    # rescale [-1,1] -> [0,W-1], +1 for the one-pixel zero pad
    u = (u + dtype(1)) * dtype(W - 1) / dtype(2) + dtype(1)
    v = (v + dtype(1)) * dtype(H - 1) / dtype(2) + dtype(1)
    uc = np.clip(u, 0, W + 1)
    vc = np.clip(v, 0, H + 1)
    with np.errstate(invalid="ignore"):
        u0 = np.clip(np.floor(np.nan_to_num(uc, nan=0.0)), 0, W).astype(np.int32)
        v0 = np.clip(np.floor(np.nan_to_num(vc, nan=0.0)), 0, H).astype(np.int32)
    u1 = u0 + 1
    v1 = v0 + 1
    wx0 = (u1 - uc).astype(dtype)
    wx1 = (uc - u0).astype(dtype)
    wy0 = (v1 - vc).astype(dtype)
    wy1 = (vc - v0).astype(dtype)
    return u, v, u0, v0, u1, v1, wx0, wx1, wy0, wy1


def _gather(xpad, v, u):
    """xpad (B,C,Hp,Wp); v,u (B,P) int -> (B,P,C)"""
    B = xpad.shape[0]
    bi = np.arange(B)[:, None]
    return xpad[bi, :, v, u]


def spatial_transformer_sampler(x, grid, dtype=np.float32):
    """Bilinear sampling at a normalized grid, zero outside the image.
    x (B,C,H,W), grid (B,2,oH,oW) with [:,0]=x in [-1,1], [:,1]=y  ->  (B,C,oH,oW)"""
    x = np.asarray(x, dtype=dtype)
    grid = np.asarray(grid, dtype=dtype)
    B, C, H, W = x.shape
    oH, oW = grid.shape[2:]
    _, _, u0, v0, u1, v1, wx0, wx1, wy0, wy1 = _sampler_coords(grid, H, W, dtype)
    xpad = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), mode="constant")
    y = (wx0 * wy0)[:, :, None] * _gather(xpad, v0, u0)
    y = y + (wx1 * wy0)[:, :, None] * _gather(xpad, v0, u1)
    y = y + (wx0 * wy1)[:, :, None] * _gather(xpad, v1, u0)
    y = y + (wx1 * wy1)[:, :, None] * _gather(xpad, v1, u1)
    return y.reshape(B, oH, oW, C).transpose(0, 3, 1, 2).astype(dtype)


def spatial_transformer_sampler_backward(x, grid, gy, dtype=np.float32, want_gx=True):
    """-> (gx (B,C,H,W) | None, ggrid (B,2,oH,oW))"""
    x = np.asarray(x, dtype=dtype)
    grid = np.asarray(grid, dtype=dtype)
    gy = np.asarray(gy, dtype=dtype)
    B, C, H, W = x.shape
    oH, oW = grid.shape[2:]
    u, v, u0, v0, u1, v1, wx0, wx1, wy0, wy1 = _sampler_coords(grid, H, W, dtype)
    xpad = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), mode="constant")
    x1 = _gather(xpad, v0, u0)
    x2 = _gather(xpad, v0, u1)
    x3 = _gather(xpad, v1, u0)
    x4 = _gather(xpad, v1, u1)
    g = gy.reshape(B, C, -1).transpose(0, 2, 1)                  # (B,P,C)
    gu = ((-wy0[:, :, None] * x1 + wy0[:, :, None] * x2 - wy1[:, :, None] * x3 + wy1[:, :, None] * x4) * g).sum(2)
    gv = ((-wx0[:, :, None] * x1 - wx1[:, :, None] * x2 + wx0[:, :, None] * x3 + wx1[:, :, None] * x4) * g).sum(2)
    with np.errstate(invalid="ignore"):
        gu = gu * dtype((W - 1) / 2.) * ((u >= 0) & (u <= W + 1))
        gv = gv * dtype((H - 1) / 2.) * ((v >= 0) & (v <= H + 1))
    ggrid = np.stack([gu, gv], axis=1).reshape(B, 2, oH, oW).astype(dtype)
    gx = None
    if want_gx:
        gxp = np.zeros_like(xpad)
        bi = np.broadcast_to(np.arange(B)[:, None], u0.shape)
        for (vv, uu, ww) in ((v0, u0, wx0 * wy0), (v0, u1, wx1 * wy0), (v1, u0, wx0 * wy1), (v1, u1, wx1 * wy1)):
            contrib = ww[:, :, None] * g                         # (B,P,C)
            for c in range(C):
                np.add.at(gxp[:, c], (bi, vv, uu), contrib[:, :, c])
        gx = gxp[:, :, 1:-1, 1:-1].astype(dtype)
    return gx, ggrid


# --------------------------------------------------------------------------------------
# the repo-local alternative sampler (spational_transformer_sampler_interp.py:32-149)
# --------------------------------------------------------------------------------------
def _interp_parts(grid, H, W, dtype):
    u = grid[:, 0].reshape(-1)                                   # :38
    v = grid[:, 1].reshape(-1)                                   # :39
    u0 = np.floor(u)                                             # :41
    u1 = u0 + 1
    v0 = np.floor(v)
    v1 = v0 + 1
    u0 = u0.clip(0, W - 1)                                       # :46-49
    v0 = v0.clip(0, H - 1)
    u1 = u1.clip(0, W - 1)
    v1 = v1.clip(0, H - 1)
    wt_x0 = u1 - u                                               # :52-55
    wt_x1 = u - u0
    wt_y0 = v1 - v
    wt_y1 = v - v0
    return (u0.astype(np.int32), v0.astype(np.int32), u1.astype(np.int32), v1.astype(np.int32),
            wt_x0.astype(dtype), wt_x1.astype(dtype), wt_y0.astype(dtype), wt_y1.astype(dtype))


def interp_sampler_forward(x, grid, dtype=np.float32):
    """SpatialTransformerSamplerInterp._forward (:32-78).  grid is in PIXEL coordinates."""
    x = np.asarray(x, dtype=dtype)
    grid = np.asarray(grid, dtype=dtype)
    B, C, H, W = x.shape
    oH, oW = grid.shape[2:]
    u0, v0, u1, v1, wx0, wx1, wy0, wy1 = _interp_parts(grid, H, W, dtype)
    bi = np.repeat(np.arange(B), oH * oW)                        # :71
    y = (wx0 * wy0)[:, None] * x[bi, :, v0, u0]                  # :57,72
    y += (wx1 * wy0)[:, None] * x[bi, :, v0, u1]                 # :58,73
    y += (wx0 * wy1)[:, None] * x[bi, :, v1, u0]                 # :59,74
    y += (wx1 * wy1)[:, None] * x[bi, :, v1, u1]                 # :60,75
    return y.reshape(B, oH, oW, C).transpose(0, 3, 1, 2)         # :77


def interp_sampler_backward(x, grid, gy, dtype=np.float32):
    """SpatialTransformerSamplerInterp._backward (:86-149) -> (gx == 0, ggrid)."""
    x = np.asarray(x, dtype=dtype)
    grid = np.asarray(grid, dtype=dtype)
    gy = np.asarray(gy, dtype=dtype)
    B, C, H, W = x.shape
    oH, oW = grid.shape[2:]
    u0, v0, u1, v1, wx0, wx1, wy0, wy1 = _interp_parts(grid, H, W, dtype)
    bi = np.repeat(np.arange(B), oH * oW)
    x1 = x[bi, :, v0, u0]
    x2 = x[bi, :, v0, u1]
    x3 = x[bi, :, v1, u0]
    x4 = x[bi, :, v1, u1]
    gu = -wy0[:, None] * x1                                      # :129-132
    gu += wy0[:, None] * x2
    gu -= wy1[:, None] * x3
    gu += wy1[:, None] * x4
    gv = -wx0[:, None] * x1                                      # :134-137
    gv -= wx1[:, None] * x2
    gv += wx0[:, None] * x3
    gv += wx1[:, None] * x4
    gu = gu.reshape(B, oH, oW, C).transpose(0, 3, 1, 2)
    gv = gv.reshape(B, oH, oW, C).transpose(0, 3, 1, 2)
    gu = (gu * gy).sum(axis=1)                                   # :142-145
    gv = (gv * gy).sum(axis=1)
    ggrid = np.concatenate((gu[:, None], gv[:, None]), axis=1)   # :147
    return np.zeros_like(x), ggrid                               # :148


# --------------------------------------------------------------------------------------
# projective_inverse_warp (transform.py:156-193) + hand-derived backward
# --------------------------------------------------------------------------------------
def projective_inverse_warp(imgs, depthes, poses, K, dtype=np.float32, return_aux=False):
    """imgs (N,3,H,W); depthes (N,3,H*W); poses (N,6); K (N,3,3) -> warped (N,3,H,W)"""
    imgs = np.asarray(imgs, dtype=dtype)
    depthes = np.asarray(depthes, dtype=dtype)
    N, _, H, W = imgs.shape
    Pm = proj_tgt_to_src(poses, K, dtype)                        # :171
    pix = generate_2dmeshgrid(H, W, N, dtype)                    # :176
    cam, ray = pixel2cam(depthes, pix, K, dtype)                 # :180
    grid, aux = cam2pixel(cam, Pm, H, W, dtype)                  # :184
    out = spatial_transformer_sampler(imgs, grid, dtype)         # :189
    if return_aux:
        # distance of the sampling position to the nearest cell boundary of the bilinear lattice
        # (where dI^/du jumps), for gradient parity bookkeeping in tests; not part of the reference
        _, _, _, _, _, _, wx0, wx1, wy0, wy1 = _sampler_coords(grid, H, W, dtype)
        aux["cell_margin"] = np.minimum(np.minimum(wx0, wx1), np.minimum(wy0, wy1)).reshape(N, H, W)
        aux.update(Pm=Pm, cam=cam, ray=ray, grid=grid)
        return out, aux
    return out


def projective_inverse_warp_backward(imgs, depthes, poses, K, g_out, dtype=np.float32, want_gimgs=False):
    """Backward of projective_inverse_warp for an upstream gradient g_out (N,3,H,W).
    -> (g_depthes (N,3,P), g_poses (N,6), g_imgs | None)"""
    imgs = np.asarray(imgs, dtype=dtype)
    depthes = np.asarray(depthes, dtype=dtype)
    N, _, H, W = imgs.shape
    P = H * W
    _, aux = projective_inverse_warp(imgs, depthes, poses, K, dtype, return_aux=True)
    gimgs, ggrid = spatial_transformer_sampler_backward(imgs, aux["grid"], g_out, dtype, want_gx=want_gimgs)
    gp = ggrid.reshape(N, 2, P) * aux["mask"]                    # p_s_xy *= mask  (transform.py:131)
    z = aux["z"]
    gU = gp[:, 0:1] / dtype((W - 1) / 2.)
    gV = gp[:, 1:2] / dtype((H - 1) / 2.)
    gq0 = gU / z
    gq1 = gV / z
    gq2 = -(gU * aux["U"] + gV * aux["V"]) / z
    gq = np.concatenate([gq0, gq1, gq2, np.zeros_like(gq0)], axis=1)      # (N,4,P)
    gPm = _bmm(gq, np.transpose(aux["cam"], (0, 2, 1)))                    # (N,4,4)
    gcam = _bmm(np.transpose(aux["Pm"], (0, 2, 1)), gq)                    # (N,4,P)
    g_depthes = gcam[:, :3] * aux["ray"]
    g_poses = proj_tgt_to_src_backward(poses, K, gPm, dtype)
    return g_depthes.astype(dtype), g_poses, gimgs


# --------------------------------------------------------------------------------------
# Chainer ops used by the loss loop -- restated, [recalled]
# --------------------------------------------------------------------------------------
def resize_images(x, out_hw, dtype=np.float32):
    """F.resize_images (base_model.py:71-72): bilinear, align-corners."""
    x = np.asarray(x, dtype=dtype)
    B, C, H, W = x.shape
    oh, ow = out_hw
    u = np.linspace(0, W - 1, num=ow).astype(dtype)
    v = np.linspace(0, H - 1, num=oh).astype(dtype)
    u0 = np.clip(np.floor(u).astype(np.int32), 0, max(W - 2, 0))
    v0 = np.clip(np.floor(v).astype(np.int32), 0, max(H - 2, 0))
    u1 = np.minimum(u0 + 1, W - 1)
    v1 = np.minimum(v0 + 1, H - 1)
    wu1 = (u - u0).astype(dtype)
    wv1 = (v - v0).astype(dtype)
    wu0 = dtype(1) - wu1
    wv0 = dtype(1) - wv1
    top = x[:, :, v0][:, :, :, u0] * wu0 + x[:, :, v0][:, :, :, u1] * wu1
    bot = x[:, :, v1][:, :, :, u0] * wu0 + x[:, :, v1][:, :, :, u1] * wu1
    return (top * wv0[:, None] + bot * wv1[:, None]).astype(dtype)


def average_pooling_3x3(x):
    """F.average_pooling_2d(x, 3, 1, 1): zero pad 1, 3x3 sum, divide by 9 always."""
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), mode="constant")
    H, W = x.shape[2:]
    acc = np.zeros_like(x)
    for dy in range(3):
        for dx in range(3):
            acc = acc + xp[:, :, dy:dy + H, dx:dx + W]
    return acc / x.dtype.type(9)


def compute_ssim(x, y, dtype=np.float32, return_aux=False):
    """base_model.py:126-142: clip((1-SSIM)/2, 0, 1)."""
    x = np.asarray(x, dtype=dtype)
    y = np.asarray(y, dtype=dtype)
    c1 = dtype(0.01 ** 2)
    c2 = dtype(0.03 ** 2)
    mu_x = average_pooling_3x3(x)                                # :130
    mu_y = average_pooling_3x3(y)                                # :131
    sigma_x = average_pooling_3x3(x ** 2) - mu_x ** 2            # :133
    sigma_y = average_pooling_3x3(y ** 2) - mu_y ** 2            # :134
    sigma_xy = average_pooling_3x3(x * y) - mu_x * mu_y          # :135
    n1 = 2 * mu_x * mu_y + c1
    n2 = 2 * sigma_xy + c2
    d1 = mu_x ** 2 + mu_y ** 2 + c1
    d2 = sigma_x + sigma_y + c2
    S = (n1 * n2) / (d1 * d2)                                    # :137-140
    e = (1 - S) / 2
    out = np.clip(e, dtype(0), dtype(1))                         # :142
    if return_aux:
        return out, dict(mu_x=mu_x, mu_y=mu_y, n1=n1, n2=n2, d1=d1, d2=d2, S=S, e=e)
    return out


def compute_ssim_backward(x, y, g, dtype=np.float32):
    """d/dx of sum(g * compute_ssim(x, y)); y-side statistics are constants (.data at :131,:134)."""
    x = np.asarray(x, dtype=dtype)
    y = np.asarray(y, dtype=dtype)
    _, a = compute_ssim(x, y, dtype, return_aux=True)
    n1, n2, d1, d2, S, e = a["n1"], a["n2"], a["d1"], a["d2"], a["S"], a["e"]
    mu_x, mu_y = a["mu_x"], a["mu_y"]
    d = d1 * d2
    kappa = g * dtype(-0.5) * ((e > 0) & (e < 1))               # F.clip backward, (1-S)/2
    dS_dExx = -S / d2
    dS_dExy = 2 * n1 / d
    dS_dmux = (2 * mu_y * (n2 - n1) - S * 2 * mu_x * (d2 - d1)) / d
    A = average_pooling_3x3(kappa * dS_dmux)                    # pool^T == pool (zero-padded 3x3 /9)
    Bq = average_pooling_3x3(kappa * dS_dExx)
    E = average_pooling_3x3(kappa * dS_dExy)
    return (A + 2 * x * Bq + y * E).astype(dtype)


def compute_smooth_loss(d, dtype=np.float32):
    """base_model.py:169-185.  d: (N,1,h,w) disparity."""
    d = np.asarray(d, dtype=dtype)

    def gradient(p):
        return p[:, :, :, 1:] - p[:, :, :, :-1], p[:, :, 1:] - p[:, :, :-1]   # (D_dx, D_dy) :176-179

    dx, dy = gradient(d)
    dx2, dxdy = gradient(dx)
    dydx, dy2 = gradient(dy)
    return sum(_mean(np.abs(t)) for t in (dx2, dxdy, dydx, dy2))


def compute_smooth_loss_backward(d, dtype=np.float32):
    """Gradient of compute_smooth_loss w.r.t. d (upstream gradient 1)."""
    d = np.asarray(d, dtype=dtype)
    g = np.zeros_like(d)

    def sgn(t):
        n = t.size
        return (np.sign(t) / dtype(n)).astype(dtype) if n else t

    dx = d[:, :, :, 1:] - d[:, :, :, :-1]
    dy = d[:, :, 1:] - d[:, :, :-1]
    gdx = np.zeros_like(dx)
    gdy = np.zeros_like(dy)
    s = sgn(dx[:, :, :, 1:] - dx[:, :, :, :-1])                  # dx2
    gdx[:, :, :, 1:] += s
    gdx[:, :, :, :-1] -= s
    s = sgn(dx[:, :, 1:] - dx[:, :, :-1])                        # dxdy
    gdx[:, :, 1:] += s
    gdx[:, :, :-1] -= s
    s = sgn(dy[:, :, :, 1:] - dy[:, :, :, :-1])                  # dydx
    gdy[:, :, :, 1:] += s
    gdy[:, :, :, :-1] -= s
    s = sgn(dy[:, :, 1:] - dy[:, :, :-1])                        # dy2
    gdy[:, :, 1:] += s
    gdy[:, :, :-1] -= s
    g[:, :, :, 1:] += gdx
    g[:, :, :, :-1] -= gdx
    g[:, :, 1:] += gdy
    g[:, :, :-1] -= gdy
    return g


def compute_disp_smooth(img, d, dtype=np.float32):
    """base_model.py:144-155 (edge-aware, dead code in the reference's loop :78-80)."""
    img = np.asarray(img, dtype=dtype)
    d = np.asarray(d, dtype=dtype)
    i_dy = img[:, :, 1:] - img[:, :, :-1]
    i_dx = img[:, :, :, 1:] - img[:, :, :, :-1]
    i_dx = i_dx.mean(axis=1, keepdims=True, dtype=np.float64).astype(dtype)
    i_dy = i_dy.mean(axis=1, keepdims=True, dtype=np.float64).astype(dtype)
    d_dy = d[:, :, 1:] - d[:, :, :-1]
    d_dx = d[:, :, :, 1:] - d[:, :, :, :-1]
    return _mean(np.abs(d_dx) * np.exp(-np.abs(i_dx))) + _mean(np.abs(d_dy) * np.exp(-np.abs(i_dy)))


def compute_disp_smooth_backward(img, d, dtype=np.float32):
    img = np.asarray(img, dtype=dtype)
    d = np.asarray(d, dtype=dtype)
    g = np.zeros_like(d)
    i_dy = (img[:, :, 1:] - img[:, :, :-1]).mean(axis=1, keepdims=True, dtype=np.float64).astype(dtype)
    i_dx = (img[:, :, :, 1:] - img[:, :, :, :-1]).mean(axis=1, keepdims=True, dtype=np.float64).astype(dtype)
    d_dy = d[:, :, 1:] - d[:, :, :-1]
    d_dx = d[:, :, :, 1:] - d[:, :, :, :-1]
    if d_dx.size:
        gx = np.sign(d_dx) * np.exp(-np.abs(i_dx)) / dtype(d_dx.size)
        g[:, :, :, 1:] += gx
        g[:, :, :, :-1] -= gx
    if d_dy.size:
        gy = np.sign(d_dy) * np.exp(-np.abs(i_dy)) / dtype(d_dy.size)
        g[:, :, 1:] += gy
        g[:, :, :-1] -= gy
    return g.astype(dtype)


def compute_exp_reg_loss(logits, dtype=np.float32):
    """base_model.py:157-167: mean sigmoid-cross-entropy against all-ones = mean softplus(-x)."""
    x = np.asarray(logits, dtype=dtype)
    return _mean(np.logaddexp(dtype(0), -x))


def get_multi_scale_intrinsics(K, n_scales, dtype=np.float32):
    """kitti_raw_transformed.py:76-93.  K (3,3) or (B,3,3) -> (..., n_scales, 3, 3)"""
    K = np.asarray(K, dtype=dtype)
    outs = []
    for s in range(n_scales):
        Ks = np.zeros_like(K)
        Ks[..., 0, 0] = K[..., 0, 0] / dtype(2 ** s)
        Ks[..., 1, 1] = K[..., 1, 1] / dtype(2 ** s)
        Ks[..., 0, 2] = K[..., 0, 2] / dtype(2 ** s)
        Ks[..., 1, 2] = K[..., 1, 2] / dtype(2 ** s)
        Ks[..., 2, 2] = 1
        outs.append(Ks)
    return np.stack(outs, axis=-3)


def _mean(a):
    """F.mean: sum / element count.  Accumulated in fp64, returned as a Python float."""
    return float(np.sum(a, dtype=np.float64) / a.size) if a.size else float("nan")


# --------------------------------------------------------------------------------------
# the loss loop (base_model.py:57-124) with hand-derived backward
# --------------------------------------------------------------------------------------
class SfmLossResult(dict):
    """total/pixel/smooth/exp/ssim losses (Python floats) + optional gradients/intermediates."""
    __getattr__ = dict.__getitem__


def sfm_loss(tgt_pyr, src_pyr, intrinsics, disps, poses, masks=None, *,
             smooth_reg=0.0, exp_reg=0.0, ssim_rate=0.0, smooth_mode="second_order",
             n_scales=None, dtype=np.float32, backward=False, want_d_src=False,
             keep_warped=False, norm_batch=None):
    """SFMLearner.__call__ from the pyramid onwards (base_model.py:69-124).

    tgt_pyr[s]   (B,3,h_s,w_s)     -- curr_tgt_img  (:71)
    src_pyr[s]   (B,3n,h_s,w_s)    -- curr_src_imgs (:72)
    intrinsics   (B,S,3,3)         -- :85
    disps[s]     (B,1,h_s,w_s)     -- pred_disps (:59); depth = 1/disp (:60)
    poses[i]     (B,6)             -- pred_poses (:62)
    masks[s]     (B,n,h_s,w_s)|None-- explainability logits (:62,:87)
    norm_batch   batch size used in the means (for a batch shard: the GLOBAL batch); default B.

    backward=True also returns d_disps[s], d_poses[i], d_masks[s] (and d_src[s] if asked)
    for an upstream gradient of 1 on total_loss.
    """
    S = len(disps) if n_scales is None else n_scales
    B = tgt_pyr[0].shape[0]
    n_src = src_pyr[0].shape[1] // 3
    Bn = B if norm_batch is None else norm_batch
    alpha = dtype(ssim_rate if ssim_rate else 0.0)
    do_exp = bool(exp_reg)
    pixel_loss = ssim_loss = smooth_loss = exp_loss = 0.0
    res = SfmLossResult()
    d_disps = [np.zeros_like(np.asarray(disps[s], dtype=dtype)) for s in range(S)]
    d_poses = [np.zeros((B, 6), dtype=dtype) for _ in range(n_src)]
    d_masks = [np.zeros_like(np.asarray(masks[s], dtype=dtype)) for s in range(S)] if do_exp else None
    d_srcs = [np.zeros_like(np.asarray(src_pyr[s], dtype=dtype)) for s in range(S)] if want_d_src else None
    warped_all, margin_all, cell_all, abs_all, clip_all, uv_all, zero_all = [], [], [], [], [], [], []

    for s in range(S):
        tgt = np.asarray(tgt_pyr[s], dtype=dtype)
        src = np.asarray(src_pyr[s], dtype=dtype)
        disp = np.asarray(disps[s], dtype=dtype)
        h, w = tgt.shape[2:]
        P = h * w
        cnt = dtype(Bn * 3 * h * w)
        scale_b = float(B) / float(Bn)       # a shard's mean over its own B, rescaled to the global batch
        if smooth_reg:                                                         # :75-77
            wgt = smooth_reg / (2 ** s)
            if smooth_mode == "second_order":
                smooth_loss += wgt * compute_smooth_loss(disp, dtype) * scale_b
                if backward:
                    d_disps[s] += dtype(wgt * scale_b) * compute_smooth_loss_backward(disp, dtype)
            elif smooth_mode == "edge_aware":                                  # :78-80 (commented out there)
                smooth_loss += wgt * compute_disp_smooth(tgt, disp, dtype) * scale_b
                if backward:
                    d_disps[s] += dtype(wgt * scale_b) * compute_disp_smooth_backward(tgt, disp, dtype)
            else:
                raise ValueError(smooth_mode)
        depth = dtype(1) / disp                                                # :60
        depthes = np.broadcast_to(depth.reshape(B, 1, P), (B, 3, P))           # :82-84
        K = np.asarray(intrinsics, dtype=dtype)[:, s]                          # :85
        w_s, m_s, c_s, a_s, k_s, uv_s, z_s = [], [], [], [], [], [], []
        for i in range(n_src):                                                 # :88
            img = src[:, 3 * i:3 * i + 3]
            proj, aux = projective_inverse_warp(img, depthes, poses[i], K, dtype, return_aux=True)   # :90-94
            err = np.abs(proj - tgt)                                           # :95
            m = (proj == 0).prod(1, keepdims=True).astype(bool)                # :96
            mb = np.broadcast_to(m, err.shape)                                 # :97
            err = np.where(mb, dtype(0), err)                                  # :98-100
            g_proj = None
            if do_exp:                                                         # :103-109
                logit = np.asarray(masks[s], dtype=dtype)[:, i:i + 1]
                exp_loss += exp_reg * compute_exp_reg_loss(logit, dtype) * scale_b
                sig = dtype(1) / (dtype(1) + np.exp(-logit))
                pixel_loss += float(np.sum(err * sig, dtype=np.float64) / cnt)
                if backward:
                    # total = (1-alpha)*pixel_loss + ... (:117); pixel term = mean(err * sigmoid(logit))
                    g_proj = (dtype(1) - alpha) * np.where(mb, dtype(0), np.sign(proj - tgt)) * sig / cnt
                    g_sig = (dtype(1) - alpha) * err.sum(axis=1, keepdims=True) / cnt
                    n_log = dtype(Bn * h * w)
                    d_masks[s][:, i:i + 1] += g_sig * sig * (1 - sig) + dtype(exp_reg) * (sig - 1) / n_log
            else:
                pixel_loss += float(np.sum(err, dtype=np.float64) / cnt)       # :111
                if backward:
                    g_proj = (dtype(1) - alpha) * np.where(mb, dtype(0), np.sign(proj - tgt)) / cnt
                if ssim_rate:                                                  # :112-115
                    se_raw, ssim_aux = compute_ssim(proj, tgt, dtype, return_aux=True)
                    e_raw = ssim_aux["e"]
                    # distance of (1-SSIM)/2 to the kinks of F.clip at 0 and 1 (test bookkeeping only)
                    # (an exact 0 -- identical flat windows -- is not a knife edge: all partials vanish there)
                    clip_m = np.minimum(np.where(e_raw == 0, dtype(1), np.abs(e_raw)), np.abs(1 - e_raw)).min(axis=1)
                    se = se_raw * (dtype(1) - mb.astype(dtype))
                    ssim_loss += float(np.sum(se, dtype=np.float64) / cnt)
                    if backward:
                        g_se = alpha * (dtype(1) - mb.astype(dtype)) / cnt
                        g_proj = g_proj + compute_ssim_backward(proj, tgt, g_se, dtype)
            if backward:
                g_dep, g_pose, g_img = projective_inverse_warp_backward(
                    img, depthes, poses[i], K, g_proj, dtype, want_gimgs=want_d_src)
                gD = g_dep.sum(axis=1).reshape(B, 1, h, w)                     # broadcast_to backward
                d_disps[s] += -gD / (disp * disp)                              # 1/d backward (:60)
                d_poses[i] += g_pose
                if want_d_src:
                    d_srcs[s][:, 3 * i:3 * i + 3] += g_img
            if keep_warped:
                w_s.append(proj)
                m_s.append(aux["margin"])
                c_s.append(np.where(m[:, 0], dtype(1), aux["cell_margin"]))
                k_s.append(clip_m if (ssim_rate and not do_exp) else np.ones_like(aux["margin"]))
                ad = np.abs(proj - tgt)
                a_s.append(np.where(m[:, 0], dtype(1), np.where(ad > 0, ad, dtype(1)).min(axis=1)))
                z_s.append(~m[:, 0] & (ad == 0).any(axis=1))      # a channel with I^ == I exactly: sign(0) = 0 here, +-1 one ulp away
                uv_s.append(np.concatenate([aux["U"], aux["V"]], axis=1).reshape(B, 2, h, w))
        if keep_warped:
            warped_all.append(np.stack(w_s, axis=1))       # (B,n,3,h,w)
            margin_all.append(np.stack(m_s, axis=1))       # (B,n,h,w)
            cell_all.append(np.stack(c_s, axis=1))         # (B,n,h,w)
            abs_all.append(np.stack(a_s, axis=1))          # (B,n,h,w)  smallest non-zero |I^ - I| (kink of |.|)
            zero_all.append(np.stack(z_s, axis=1))         # (B,n,h,w)  in view and I^ == I exactly in some channel (ON the kink)
            clip_all.append(np.stack(k_s, axis=1))         # (B,n,h,w)  distance of (1-SSIM)/2 to the clip kinks
            uv_all.append(np.stack(uv_s, axis=1))          # (B,n,2,h,w) sampling position (U, V) in source pixels

    total = (1 - float(alpha)) * pixel_loss + float(alpha) * ssim_loss + smooth_loss + exp_loss   # :117-118
    res.update(total_loss=total, pixel_loss=pixel_loss, smooth_loss=smooth_loss,
               exp_loss=exp_loss, ssim_loss=ssim_loss)
    if backward:
        res.update(d_disps=d_disps, d_poses=d_poses, d_masks=d_masks, d_srcs=d_srcs)
    if keep_warped:
        res.update(warped=warped_all, margin=margin_all, cell_margin=cell_all, abs_margin=abs_all, abs_zero=zero_all, clip_margin=clip_all, uv=uv_all)
    return res


# --------------------------------------------------------------------------------------
# data augmentation (datasets/kitti/kitti_raw_transformed.py:23-74), one sample, explicit parameters
# --------------------------------------------------------------------------------------
def data_augmentation(tgt_img, src_imgs, intrinsics, x_scaling, y_scaling, offset_y, offset_x, flip, dtype=np.float32):
    """tgt (3,H,W), src (S,3,H,W), K (3,3) with the random draws of :34,:50-51,:64 passed in.
    -> (tgt, src, K) exactly as the reference composes them: resize -> crop -> flip."""
    _, out_h, out_w = tgt_img.shape
    imgs = np.concatenate((tgt_img[np.newaxis, :], src_imgs)).astype(dtype)        # :70
    K = np.asarray(intrinsics, dtype='f')
    in_h, in_w = imgs.shape[2:]
    sh, sw = int(in_h * y_scaling), int(in_w * x_scaling)                           # :37-38
    imgs = resize_images(imgs, (sh, sw), dtype)                                     # :39
    K = np.array([[K[0, 0] * x_scaling, 0., K[0, 2] * x_scaling],
                  [0., K[1, 1] * y_scaling, K[1, 2] * y_scaling], [0., 0., 1.]], dtype='f')   # :40-44
    imgs = imgs[:, :, offset_y:offset_y + out_h, offset_x:offset_x + out_w]         # :52
    K = np.array([[K[0, 0], 0., K[0, 2] - offset_x], [0., K[1, 1], K[1, 2] - offset_y], [0., 0., 1.]], dtype='f')   # :53-58
    if flip:                                                                         # :64-66
        imgs = imgs[:, :, :, ::-1]
        K[0, 2] = imgs.shape[3] - K[0, 2]
    return imgs[0], imgs[1:], K
