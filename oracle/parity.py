"""Parity bookkeeping shared by tests/ and __graft_entry__.smoke() -- TEST INFRASTRUCTURE, like the rest of oracle/: never
imported by the product path.

The reference's function is discontinuous in (disp, pose) at a few kinds of pixels; two correct fp32 evaluations may land on
different sides there.  `knife_mask` names those pixels from the ORACLE's own margins (sfm_oracle.sfm_loss(keep_warped=True)),
with the footprint each kind can influence; gradient comparisons are element-wise outside the mask."""
import numpy as np


def dilate(mask, r):
    """Binary dilation of the last two axes by a (2r+1)^2 box."""
    out = mask.copy()
    H, W = mask.shape[-2:]
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            ys = slice(max(0, dy), H + min(0, dy))
            yd = slice(max(0, -dy), H + min(0, -dy))
            xs = slice(max(0, dx), W + min(0, dx))
            xd = slice(max(0, -dx), W + min(0, -dx))
            out[..., yd, xd] |= mask[..., ys, xs]
    return out


def knife_mask(ref, s, thr=8e-6, cell_thr=1e-4, abs_thr=3e-5, clip_thr=5e-5):
    """The knife-edge pixels of scale s with their footprints, (B,h,w) bool, and the three classes before dilation:
      * flip: the strict `-1 < x < 1` test (models/transform.py:129) within `thr` of its boundary -- the pixel flips between
        sampled and exactly 0, which changes the SSIM windows around it: 5x5 footprint;
      * clip: (1-SSIM)/2 within `clip_thr` of the kinks of F.clip at 0 / 1 (models/base_model.py:142): 3x3 footprint;
      * own: the sample within `cell_thr` px of a cell boundary of the bilinear lattice (dI^/du jumps) or 0 < |I^ - I| < `abs_thr`
        (kink of F.absolute, models/base_model.py:95): the pixel itself."""
    flip = (ref["margin"][s] < thr).any(axis=1)
    clip = (ref["clip_margin"][s] < clip_thr).any(axis=1)
    own = (ref["cell_margin"][s] < cell_thr).any(axis=1) | (ref["abs_margin"][s] < abs_thr).any(axis=1)
    return dilate(flip, 2) | dilate(clip, 1) | own, flip, clip, own


def rel_l2(got, want, knife=None):
    """Relative L2 error of `got` against `want` outside the (broadcastable) mask `knife`."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    keep = np.ones(got.shape, bool) if knife is None else ~np.broadcast_to(knife, got.shape)
    return float(np.sqrt((((got - want) * keep) ** 2).sum()) / max(np.sqrt(((want * keep) ** 2).sum()), 1e-30))


def position_uncertainty(K, pose, disp, k=2.0):
    """How far two correct fp32 evaluations of the sampling position (U, V) = (q0/z, q1/z) of models/transform.py:105-125 may lie
    apart: a first-order running error bound, evaluated in fp64 from the INPUTS alone (no implementation's rounding enters).
    Every sum of the chain  ray = K^-1 pix,  cam = D ray,  q = Pm (cam, 1)  is charged k unit roundoffs (2^-24) of the sum of the
    magnitudes of its terms (cancellation is what makes a position uncertain: q0 = fx X + cx Z sums to D x from terms of size
    D (x + 2 cx)), and U = q0 / z inherits  (E_q0 + |U| E_z) / |z|  -- which grows without bound where z -> 0.
      K (B,3,3), pose (B,6), disp (B,1,h,w)  ->  dU, dV (B,h,w) in source pixels.
    At a 128x416 BASELINE frame with small motion: 0.5e-4 .. 2e-4 px (1.5 .. 7 ulps of U); the measured differences between the
    fp32 oracle and the fp64 oracle reach 1.0e-4 px there."""
    from . import sfm_oracle as O
    K = np.asarray(K, np.float64)
    B = K.shape[0]
    h, w = disp.shape[2:]
    Pm = O.proj_tgt_to_src(np.asarray(pose, np.float64), K, np.float64)[:, :3]          # (B,3,4)
    Kinv = np.linalg.inv(K)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    pix = np.stack([xs, ys, np.ones_like(xs)])                                         # (3,h,w)
    D = 1.0 / np.asarray(disp, np.float64)[:, 0]                                       # (B,h,w)
    cam = D[:, None] * np.einsum("bij,jhw->bihw", Kinv, pix)
    cam_abs = D[:, None] * np.einsum("bij,jhw->bihw", np.abs(Kinv), pix)
    q = np.einsum("bij,bjhw->bihw", Pm[:, :, :3], cam) + Pm[:, :, 3, None, None]
    E = np.einsum("bij,bjhw->bihw", np.abs(Pm[:, :, :3]), cam_abs) + np.abs(Pm[:, :, 3, None, None])
    z = q[:, 2] + 1e-10
    with np.errstate(divide="ignore", invalid="ignore"):
        U, V = q[:, 0] / z, q[:, 1] / z
        g = k * 2.0 ** -24
        dU = g * (E[:, 0] + np.abs(U) * E[:, 2]) / np.abs(z)
        dV = g * (E[:, 1] + np.abs(V) * E[:, 2]) / np.abs(z)
    return np.nan_to_num(dU, nan=np.inf), np.nan_to_num(dV, nan=np.inf)


def tap_contrast(img, U, V):
    """Largest difference between horizontally / vertically adjacent taps of the bilinear cell a sample at (U, V) falls in (over
    the channels): the Lipschitz constants of the bilinear sample in u and v there.  img (B,C,h,w); U, V (B,h,w) -> Gu, Gv (B,h,w)
    (0 where the sample is not inside the image)."""
    img = np.asarray(img, np.float64)
    B, C, h, w = img.shape
    du = np.abs(img[:, :, :, 1:] - img[:, :, :, :-1]).max(axis=1)      # (B,h,w-1)
    dv = np.abs(img[:, :, 1:, :] - img[:, :, :-1, :]).max(axis=1)      # (B,h-1,w)
    with np.errstate(invalid="ignore"):
        ok = np.isfinite(U) & np.isfinite(V) & (U >= 0) & (U <= w - 1) & (V >= 0) & (V <= h - 1)
    u0 = np.clip(np.floor(np.where(ok, U, 0)).astype(np.int64), 0, w - 2)
    v0 = np.clip(np.floor(np.where(ok, V, 0)).astype(np.int64), 0, h - 2)
    bi = np.arange(B)[:, None, None]
    Gu = np.maximum(du[bi, v0, u0], du[bi, v0 + 1, u0])
    Gv = np.maximum(dv[bi, v0, u0], dv[bi, v0, u0 + 1])
    return np.where(ok, Gu, 0.0), np.where(ok, Gv, 0.0)
