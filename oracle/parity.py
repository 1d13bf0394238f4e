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
