"""Thin array-level wrappers over the C ABI.  torch.Tensor on a ROCm device is used purely as
the device-array container (allocation, stream, lifetime); all arithmetic happens in
libsfmwarp.so.  Arrays are float32, C-contiguous, NCHW -- the reference's layout.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import SfmLossDesc, check, lib

__all__ = ["pose_proj_fwd", "pose_proj_bwd", "warp_fwd", "warp_bwd", "sampler_fwd", "sampler_bwd",
           "interp_fwd", "interp_bwd", "resize", "pyramid", "disp_act_fwd", "disp_act_bwd", "FusedLoss"]


def _dev(t, name, ndim=None):
    """float32 CUDA(ROCm) tensor, contiguous; anything else is a type error, as in the
    reference's check_type_forward (spational_transformer_sampler_interp.py:11-24)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s: expected a torch.Tensor on a ROCm device, got %s" % (name, type(t).__name__))
    if not t.is_cuda:
        raise TypeError("%s: CPU arrays are not supported by this build (GPU-only, no CPU fallback)" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s: expected dtype float32 (dtype.char == 'f'), got %s" % (name, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise TypeError("%s: expected ndim == %d, got %d" % (name, ndim, t.dim()))
    return t.contiguous()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(index=None):
    """The HIP stream torch currently issues work on for that device, as a raw handle.  (The private accessor
    skips the construction of a torch.cuda.Stream object: it is what torch's own launchers use per call.)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device() if index is None else index))
    return C.c_void_p(torch.cuda.current_stream(index).cuda_stream)


def pose_proj_fwd(pose6, K):
    pose6, K = _dev(pose6, "pose6", 2), _dev(K, "K", 3)
    N = pose6.shape[0]
    if pose6.shape[1] != 6 or tuple(K.shape) != (N, 3, 3):
        raise TypeError("pose6 must be (N,6) and K (N,3,3)")
    out = torch.empty((N, 4, 4), dtype=torch.float32, device=pose6.device)
    with torch.cuda.device(pose6.device):
        check(lib.sfm_pose_proj_fwd(_p(pose6), _p(K), _p(out), N, _stream()))
    return out


def pose_proj_bwd(pose6, K, g_proj):
    pose6, K, g_proj = _dev(pose6, "pose6", 2), _dev(K, "K", 3), _dev(g_proj, "g_proj", 3)
    N = pose6.shape[0]
    out = torch.empty((N, 6), dtype=torch.float32, device=pose6.device)
    with torch.cuda.device(pose6.device):
        check(lib.sfm_pose_proj_bwd(_p(pose6), _p(K), _p(g_proj), _p(out), N, _stream()))
    return out


def _warp_args(imgs, depth, pose6, K):
    imgs = _dev(imgs, "imgs", 4)
    N, Cc, H, W = imgs.shape
    depth = _dev(depth, "depth")
    if depth.numel() == N * H * W:
        drows = 1
    elif depth.numel() == 3 * N * H * W:
        drows = 3
    else:
        raise TypeError("depthes must be (N,3,H*W) or (N,H*W), got shape %s" % (tuple(depth.shape),))
    pose6, K = _dev(pose6, "poses", 2), _dev(K, "K", 3)
    if tuple(pose6.shape) != (N, 6) or tuple(K.shape) != (N, 3, 3):
        raise TypeError("poses must be (N,6) and K (N,3,3) with N=%d" % N)
    return imgs, depth, drows, pose6, K, N, Cc, H, W


def warp_fwd(imgs, depth, pose6, K):
    """projective_inverse_warp forward.  depth: (N,3,H*W) as in the reference, or (N,H*W) = one
    row of its broadcast (models/base_model.py:82-84)."""
    imgs, depth, drows, pose6, K, N, Cc, H, W = _warp_args(imgs, depth, pose6, K)
    out = torch.empty_like(imgs)
    with torch.cuda.device(imgs.device):
        check(lib.sfm_warp_fwd(_p(imgs), _p(depth), drows, _p(pose6), _p(K), _p(out), N, Cc, H, W, _stream()))
    return out


def warp_bwd(imgs, depth, pose6, K, g_warped, want_d_src=False):
    imgs, depth, drows, pose6, K, N, Cc, H, W = _warp_args(imgs, depth, pose6, K)
    g_warped = _dev(g_warped, "g_warped", 4)
    if g_warped.shape != imgs.shape:
        raise TypeError("g_warped must have the shape of imgs")
    d_depth = torch.empty((N, 3, H * W) if drows == 3 else (N, H * W), dtype=torch.float32, device=imgs.device)
    d_pose = torch.empty((N, 6), dtype=torch.float32, device=imgs.device)
    d_src = torch.zeros_like(imgs) if want_d_src else None
    nbytes = lib.sfm_warp_bwd_workspace_bytes(N, H, W)
    ws = torch.empty((max(nbytes, 4) // 4,), dtype=torch.float32, device=imgs.device)
    with torch.cuda.device(imgs.device):
        check(lib.sfm_warp_bwd(_p(imgs), _p(depth), drows, _p(pose6), _p(K), _p(g_warped), _p(d_depth), _p(d_pose),
                               _p(d_src), _p(ws), nbytes, N, Cc, H, W, _stream()))
    return d_depth, d_pose, d_src


def _sampler_args(x, grid):
    x, grid = _dev(x, "x", 4), _dev(grid, "grid", 4)
    if grid.shape[1] != 2:
        raise TypeError("grid.shape[1] must be 2, got %d" % grid.shape[1])
    if x.shape[0] != grid.shape[0]:
        raise TypeError("x.shape[0] != grid.shape[0] (%d vs %d)" % (x.shape[0], grid.shape[0]))
    N, Cc, H, W = x.shape
    return x, grid, N, Cc, H, W, grid.shape[2], grid.shape[3]


def _sampler(fwd, x, grid):
    x, grid, N, Cc, H, W, oH, oW = _sampler_args(x, grid)
    y = torch.empty((N, Cc, oH, oW), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(fwd(_p(x), _p(grid), _p(y), N, Cc, H, W, oH, oW, _stream()))
    return y


def _sampler_b(bwd, x, grid, gy, want_gx):
    x, grid, N, Cc, H, W, oH, oW = _sampler_args(x, grid)
    gy = _dev(gy, "gy", 4)
    if tuple(gy.shape) != (N, Cc, oH, oW):
        raise TypeError("gy must be (N,C,oH,oW)")
    ggrid = torch.empty_like(grid)
    gx = torch.zeros_like(x) if want_gx else None
    with torch.cuda.device(x.device):
        check(bwd(_p(x), _p(grid), _p(gy), _p(ggrid), _p(gx), N, Cc, H, W, oH, oW, _stream()))
    return gx, ggrid


def sampler_fwd(x, grid):
    return _sampler(lib.sfm_sampler_fwd, x, grid)


def sampler_bwd(x, grid, gy, want_gx=True):
    return _sampler_b(lib.sfm_sampler_bwd, x, grid, gy, want_gx)


def interp_fwd(x, grid):
    return _sampler(lib.sfm_sampler_interp_fwd, x, grid)


def interp_bwd(x, grid, gy, want_gx=True):
    return _sampler_b(lib.sfm_sampler_interp_bwd, x, grid, gy, want_gx)


def resize(x, out_hw):
    x = _dev(x, "x", 4)
    N, Cc, H, W = x.shape
    oH, oW = int(out_hw[0]), int(out_hw[1])
    y = torch.empty((N, Cc, oH, oW), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.sfm_resize_fwd(_p(x), _p(y), N, Cc, H, W, oH, oW, _stream()))
    return y


def pyramid(x, n_scales, out=None):
    """[x, resize(x, (H>>1, W>>1)), ...]: all scales of models/base_model.py:69-72 in one launch.
    `out`: the list an earlier call with the same shapes returned -- scales 1.. are written in place and scale 0 becomes `x`
    (a caller that needs scale 0 at a fixed address copies it)."""
    x = _dev(x, "x", 4)
    N, Cc, H, W = x.shape
    if not 1 <= n_scales <= _lib.SFM_MAX_SCALES:
        raise TypeError("n_scales must be in [1, %d]" % _lib.SFM_MAX_SCALES)
    if out is not None:
        if len(out) != n_scales or any(tuple(out[s].shape) != (N, Cc, H >> s, W >> s) for s in range(n_scales)):
            raise TypeError("pyramid: `out` does not match the input")
        outs = [x] + list(out[1:])
    else:
        outs = [x] + [torch.empty((N, Cc, H >> s, W >> s), dtype=torch.float32, device=x.device) for s in range(1, n_scales)]
    ptrs = (C.c_void_p * n_scales)(*[t.data_ptr() for t in outs])
    with torch.cuda.device(x.device):
        check(lib.sfm_pyramid_fwd(_p(x), ptrs, N, Cc, H, W, n_scales, _stream()))
    return outs


def pyramid_hwc(x, n_scales):
    """The pyramid of `pyramid`, pixel-interleaved for the fused loss (SFM_LAYOUT_HWC): x (N,3G,H,W) planar, G images
    per sample -> [y_s (N,G,H>>s,W>>s,3) for s in 0..n_scales-1], one launch.  Same values as `pyramid`."""
    x = _dev(x, "x", 4)
    N, Cc, H, W = x.shape
    if Cc % 3 != 0:
        raise TypeError("pyramid_hwc: the channel count must be a multiple of 3 (RGB images), got %d" % Cc)
    if not 1 <= n_scales <= _lib.SFM_MAX_SCALES:
        raise TypeError("n_scales must be in [1, %d]" % _lib.SFM_MAX_SCALES)
    G = Cc // 3
    outs = [torch.empty((N, G, H >> s, W >> s, 3), dtype=torch.float32, device=x.device) for s in range(n_scales)]
    ptrs = (C.c_void_p * n_scales)(*[t.data_ptr() for t in outs])
    with torch.cuda.device(x.device):
        check(lib.sfm_pyramid_hwc_fwd(_p(x), ptrs, N, G, H, W, n_scales, _stream()))
    return outs


def pyramid_pair_hwc(tgt, src, n_scales, out=None, per_pixel=False):
    """Both pixel-interleaved pyramids of a step in one launch: tgt (N,3,H,W), src (N,3*n_src,H,W) ->
    ([tgt_s (N,1,h,w,3)], [src_s (N,n_src,h,w,3)]) -- the loop head models/base_model.py:69-72.
    `out` = (yt, ys) of an earlier call with the same shapes: written in place instead of allocating.
    `per_pixel`: run the one-thread-per-output-pixel kernel instead of the band kernel (same values bit for bit; A/B tests)."""
    tgt, src = _dev(tgt, "tgt", 4), _dev(src, "src", 4)
    N, Ct, H, W = tgt.shape
    if Ct != 3 or src.shape[0] != N or tuple(src.shape[2:]) != (H, W) or src.shape[1] % 3 != 0 or src.shape[1] == 0:
        raise TypeError("pyramid_pair_hwc: expected tgt (N,3,H,W) and src (N,3*n_src,H,W), got %s and %s" % (tuple(tgt.shape), tuple(src.shape)))
    if not 1 <= n_scales <= _lib.SFM_MAX_SCALES:
        raise TypeError("n_scales must be in [1, %d]" % _lib.SFM_MAX_SCALES)
    n_src = src.shape[1] // 3
    if out is not None:
        yt, ys = out
        if len(yt) != n_scales or len(ys) != n_scales or tuple(yt[0].shape) != (N, 1, H, W, 3) or tuple(ys[0].shape) != (N, n_src, H, W, 3):
            raise TypeError("pyramid_pair_hwc: `out` does not match the inputs")
        if yt[0].device != tgt.device or ys[0].device != tgt.device:
            raise TypeError("pyramid_pair_hwc: `out` lives on %s, the inputs on %s" % (yt[0].device, tgt.device))
        # (the pointer arrays of the buffers are built once -- a step is 15-60 us -- and are only reused while `out` still holds the
        #  very arrays they were built from: round-5 advisor finding, a caller that swapped an element used to be written through a
        #  stale pointer)
        key = tuple(t.data_ptr() for t in yt) + tuple(t.data_ptr() for t in ys)
        cached = getattr(out, "_ptrs", None)
        ptrs = cached[1] if cached is not None and cached[0] == key else None
        if ptrs is None:
            for s in range(n_scales):
                if tuple(yt[s].shape) != (N, 1, H >> s, W >> s, 3) or tuple(ys[s].shape) != (N, n_src, H >> s, W >> s, 3) or \
                        not yt[s].is_contiguous() or not ys[s].is_contiguous() or yt[s].dtype != torch.float32 or ys[s].dtype != torch.float32 or \
                        yt[s].device != tgt.device or ys[s].device != tgt.device:
                    raise TypeError("pyramid_pair_hwc: `out` scale %d does not match the inputs" % s)
    else:
        yt = tuple(torch.empty((N, 1, H >> s, W >> s, 3), dtype=torch.float32, device=tgt.device) for s in range(n_scales))
        ys = tuple(torch.empty((N, n_src, H >> s, W >> s, 3), dtype=torch.float32, device=tgt.device) for s in range(n_scales))
        out, ptrs = _PyramidPair((yt, ys)), None
    if ptrs is None:
        ptrs = (_ptr_array(yt), _ptr_array(ys))
        if isinstance(out, _PyramidPair):
            out._ptrs = (tuple(t.data_ptr() for t in yt) + tuple(t.data_ptr() for t in ys), ptrs)
    idx = tgt.device.index
    if torch.cuda.current_device() == idx:
        if per_pixel:
            check(lib.sfm_pyramid_variant(1))
        check(lib.sfm_pyramid_pair_hwc_fwd(_p(tgt), _p(src), ptrs[0], ptrs[1], N, n_src, H, W, n_scales, _stream(idx)))
    else:
        with torch.cuda.device(tgt.device):
            if per_pixel:
                check(lib.sfm_pyramid_variant(1))
            check(lib.sfm_pyramid_pair_hwc_fwd(_p(tgt), _p(src), ptrs[0], ptrs[1], N, n_src, H, W, n_scales, _stream(idx)))
    return out


class _PyramidPair(tuple):
    """(tgt pyramid, src pyramid) as `pyramid_pair_hwc` returns it -- two TUPLES of arrays; carries the ctypes pointer arrays of its
    buffers, keyed on their addresses, so that a caller that hands it back as `out` does not pay for rebuilding them every step."""


def to_hwc(x):
    """(N,3G,h,w) planar -> (N,G,h,w,3) pixel-interleaved copy (a torch permute; for callers that hold planar pyramids)."""
    N, Cc, h, w = x.shape
    return x.reshape(N, Cc // 3, 3, h, w).permute(0, 1, 3, 4, 2).contiguous()


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def disp_act_fwd(xs):
    """[10 * sigmoid(x) + 0.01 for x in xs] in one launch (models/disp_net.py:104-122)."""
    xs = [_dev(x, "xs[%d]" % k) for k, x in enumerate(xs)]
    outs = [torch.empty_like(x) for x in xs]
    n = (C.c_longlong * len(xs))(*[x.numel() for x in xs])
    with torch.cuda.device(xs[0].device):
        check(lib.sfm_disp_act_fwd(_ptr_array(xs), _ptr_array(outs), n, len(xs), _stream()))
    return outs


def disp_act_bwd(disps, g_disps):
    disps = [_dev(x, "disps[%d]" % k) for k, x in enumerate(disps)]
    g_disps = [_dev(x, "g_disps[%d]" % k) for k, x in enumerate(g_disps)]
    if any(a.shape != b.shape for a, b in zip(disps, g_disps)):
        raise TypeError("g_disps must match disps")
    outs = [torch.empty_like(x) for x in disps]
    n = (C.c_longlong * len(disps))(*[x.numel() for x in disps])
    with torch.cuda.device(disps[0].device):
        check(lib.sfm_disp_act_bwd(_ptr_array(disps), _ptr_array(g_disps), _ptr_array(outs), n, len(disps), _stream()))
    return outs


class FusedLoss:
    """One bound instance of the fused multi-scale loss (sfm_loss_fwd / _bwd / _fwd_bwd):
    descriptor + caller-owned workspace and outputs.  Re-usable across steps as long as the
    input tensors keep their addresses (call `bind` again otherwise)."""

    def __init__(self, smooth_reg=0.0, exp_reg=0.0, ssim_rate=0.0, smooth_mode="second_order", projection="fast"):
        """projection: "fast" (SFM_PROJECTION_FAST) or "reference_order" (SFM_PROJECTION_REFERENCE_ORDER: the per-pixel chain of
        models/transform.py:105-108,122-131 in the reference's own rounding sequence; include/sfmwarp.h says what each guarantees)."""
        if smooth_mode not in _lib.SMOOTH_MODES:
            raise ValueError("smooth_mode must be one of %s" % sorted(k for k in _lib.SMOOTH_MODES if k))
        if projection not in _lib.PROJECTIONS:
            raise ValueError("projection must be one of %s" % sorted(k for k in _lib.PROJECTIONS if k))
        self.projection = _lib.PROJECTIONS[projection]
        self.smooth_reg = float(smooth_reg or 0.0)
        self.exp_reg = float(exp_reg or 0.0)
        self.ssim_rate = float(ssim_rate or 0.0)
        self.smooth_mode = _lib.SMOOTH_MODES[smooth_mode]
        self.desc = None
        self._keep = None

    def bind(self, tgt_pyr, src_pyr, intrinsics, disps, poses, masks=None, norm_B=None, want_d_src=False, layout="planar",
             want_warped=False):
        """layout: "planar" -- tgt (B,3,h,w), src (B,3*n_src,h,w) as in the reference; "hwc" -- tgt (B,1,h,w,3),
        src (B,n_src,h,w,3) as written by `pyramid_hwc` (the faster layout for these kernels; same results).
        want_warped: `forward` / `forward_backward` also write `self.warped[s]` (B,n_src,3,h,w), the warped source images the
        loss was computed on (curr_proj_img, models/base_model.py:90-94; planar in both layouts)."""
        if layout not in ("planar", "hwc"):
            raise ValueError("layout must be 'planar' or 'hwc', got %r" % (layout,))
        hwc = layout == "hwc"
        S = len(disps)
        if not (len(tgt_pyr) == len(src_pyr) == S):
            raise TypeError("tgt_pyr, src_pyr and disps must have one entry per scale")
        if S > _lib.SFM_MAX_SCALES or len(poses) > _lib.SFM_MAX_SRC:
            raise TypeError("at most %d scales and %d sources" % (_lib.SFM_MAX_SCALES, _lib.SFM_MAX_SRC))
        tgt_pyr = [_dev(t, "tgt_pyr[%d]" % s, 5 if hwc else 4) for s, t in enumerate(tgt_pyr)]
        src_pyr = [_dev(t, "src_pyr[%d]" % s, 5 if hwc else 4) for s, t in enumerate(src_pyr)]
        disps = [_dev(t, "disps[%d]" % s, 4) for s, t in enumerate(disps)]
        poses = [_dev(t, "poses[%d]" % i, 2) for i, t in enumerate(poses)]
        intrinsics = _dev(intrinsics, "intrinsics", 4)
        B = tgt_pyr[0].shape[0]
        n_src = len(poses)
        dev = tgt_pyr[0].device
        if tuple(intrinsics.shape) != (B, S, 3, 3):
            raise TypeError("intrinsics must be (B,%d,3,3), got %s" % (S, tuple(intrinsics.shape)))
        use_masks = self.exp_reg > 0
        if use_masks:
            if masks is None:
                raise ValueError("exp_reg > 0 needs the explainability logits (masks)")
            masks = [_dev(t, "masks[%d]" % s, 4) for s, t in enumerate(masks)]
        d = SfmLossDesc()
        d.B, d.norm_B, d.n_src, d.n_scales = B, int(norm_B if norm_B is not None else B), n_src, S
        d.smooth_reg, d.exp_reg, d.ssim_rate, d.smooth_mode = self.smooth_reg, self.exp_reg, self.ssim_rate, self.smooth_mode
        d.intrinsics = intrinsics.data_ptr()
        d.image_layout = _lib.SFM_LAYOUT_HWC if hwc else _lib.SFM_LAYOUT_PLANAR
        d.projection = self.projection
        d_disps, d_masks, d_srcs, warped = [], [], [], []
        # the d_src arrays of all bound scales are views of ONE allocation: the library accumulates into them (float atomics), so every
        # backward starts by clearing them -- one fill kernel instead of one per scale
        want_src = [bool(want_d_src[s] if isinstance(want_d_src, (list, tuple)) else want_d_src) for s in range(S)]
        src_numel = [B * 3 * n_src * int(disps[s].shape[2]) * int(disps[s].shape[3]) if want_src[s] else 0 for s in range(S)]
        self._d_src_all = torch.zeros((sum(src_numel),), dtype=torch.float32, device=dev) if any(want_src) else None
        src_off = 0
        for s in range(S):
            h, w = disps[s].shape[2:]
            if hwc:
                ok = tuple(tgt_pyr[s].shape) == (B, 1, h, w, 3) and tuple(src_pyr[s].shape) == (B, n_src, h, w, 3)
            else:
                ok = tuple(tgt_pyr[s].shape) == (B, 3, h, w) and tuple(src_pyr[s].shape) == (B, 3 * n_src, h, w)
            if not ok or tuple(disps[s].shape) != (B, 1, h, w):
                raise TypeError("scale %d: expected tgt (B,3,h,w), src (B,3*n_src,h,w) [hwc: (B,1,h,w,3), (B,n_src,h,w,3)], "
                                "disp (B,1,h,w)" % s)
            d.H[s], d.W[s] = h, w
            d.tgt[s], d.src[s], d.disp[s] = tgt_pyr[s].data_ptr(), src_pyr[s].data_ptr(), disps[s].data_ptr()
            d_disps.append(torch.empty_like(disps[s]))
            d.d_disp[s] = d_disps[-1].data_ptr()
            if use_masks:
                if tuple(masks[s].shape) != (B, n_src, h, w):
                    raise TypeError("masks[%d] must be (B,n_src,h,w)" % s)
                d.mask_logits[s] = masks[s].data_ptr()
                d_masks.append(torch.empty_like(masks[s]))
                d.d_mask[s] = d_masks[-1].data_ptr()
            # (want_d_src: True, or one flag per scale -- SfmLossDesc.d_src[s] may be NULL for any scale)
            if want_src[s]:      # always planar
                d_srcs.append(self._d_src_all[src_off:src_off + src_numel[s]].view(B, 3 * n_src, h, w))
                src_off += src_numel[s]
                d.d_src[s] = d_srcs[-1].data_ptr()
            else:
                d_srcs.append(None)
            if want_warped:     # always planar
                warped.append(torch.empty((B, n_src, 3, h, w), dtype=torch.float32, device=dev))
                d.warped[s] = warped[-1].data_ptr()
        d_poses = []
        for i in range(n_src):
            if tuple(poses[i].shape) != (B, 6):
                raise TypeError("poses[%d] must be (B,6)" % i)
            d.pose[i] = poses[i].data_ptr()
            d_poses.append(torch.empty_like(poses[i]))
            d.d_pose[i] = d_poses[-1].data_ptr()
        nbytes = lib.sfm_loss_workspace_bytes(C.byref(d)) if B > 0 else 256
        if nbytes == 0:
            check(lib.sfm_loss_fwd(C.byref(d), None, None, 0, None))   # re-run the validation for its message
            raise ValueError(_lib.last_error() or "invalid loss descriptor")
        self.ws = torch.empty((nbytes // 4 + 64,), dtype=torch.float32, device=dev)
        off = (-self.ws.data_ptr()) % 256
        self._ws_ptr = self.ws.data_ptr() + off
        self._ws_bytes = nbytes
        self.loss5 = torch.zeros((5,), dtype=torch.float32, device=dev)
        self.desc, self.device = d, dev
        self._desc_ref, self._ws_arg, self._loss5_arg = C.byref(d), C.c_void_p(self._ws_ptr), _p(self.loss5)
        self.d_disps, self.d_poses, self.d_masks, self.d_srcs = d_disps, d_poses, (d_masks if use_masks else None), \
            (d_srcs if any(t is not None for t in d_srcs) else None)
        self.warped = warped if want_warped else None
        self._keep = (tgt_pyr, src_pyr, intrinsics, disps, poses, masks)
        return self

    def rebind(self, intrinsics, disps, poses, masks=None):
        """Points the bound descriptor at other input arrays of the SAME shapes (the network outputs of the next
        iteration); pyramids, workspace and gradient buffers stay.  Only pointers change: the plan cached inside the
        library for this descriptor shape is found again as long as the addresses repeat."""
        d = self.desc
        old = self._keep
        intrinsics = _dev(intrinsics, "intrinsics", 4)
        disps = [_dev(t, "disps[%d]" % s, 4) for s, t in enumerate(disps)]
        poses = [_dev(t, "poses[%d]" % i, 2) for i, t in enumerate(poses)]
        if intrinsics.shape != old[2].shape or len(disps) != len(old[3]) or len(poses) != len(old[4]) \
                or any(a.shape != b.shape for a, b in zip(disps, old[3])) or any(a.shape != b.shape for a, b in zip(poses, old[4])):
            raise TypeError("rebind: shapes differ from the bound ones (call bind)")
        if self.exp_reg > 0:
            if masks is None:
                raise ValueError("exp_reg > 0 needs the explainability logits (masks)")
            masks = [_dev(t, "masks[%d]" % s, 4) for s, t in enumerate(masks)]
            if any(a.shape != b.shape for a, b in zip(masks, old[5])):
                raise TypeError("rebind: mask shapes differ from the bound ones (call bind)")
            for s, t in enumerate(masks):
                d.mask_logits[s] = t.data_ptr()
        else:
            masks = None
        d.intrinsics = intrinsics.data_ptr()
        for s, t in enumerate(disps):
            d.disp[s] = t.data_ptr()
        for i, t in enumerate(poses):
            d.pose[i] = t.data_ptr()
        self._keep = (old[0], old[1], intrinsics, disps, poses, masks)
        return self

    def _zero_d_src(self):
        if self.d_srcs is not None:
            self._d_src_all.zero_()

    def _launch(self, fn, *mid):
        """One call through the C ABI on the device's current stream.  The argument objects that never change between
        calls are built once per bind (a step is 70 us of GPU time: the host side of a call has to stay well below)."""
        idx = self.device.index
        if torch.cuda.current_device() == idx:
            check(fn(self._desc_ref, *mid, self._ws_arg, self._ws_bytes, _stream(idx)))
        else:
            with torch.cuda.device(self.device):
                check(fn(self._desc_ref, *mid, self._ws_arg, self._ws_bytes, _stream(idx)))

    def step_from_frames(self, tgt_full, src_full, grad=True, out=None):
        """One step from the FULL-RESOLUTION frames in one call through the C ABI (sfm_step_fwd_bwd / sfm_step_fwd): both pyramids
        are written into the pixel-interleaved buffers this instance was bound to (layout="hwc"), then the fused loss runs.
        tgt_full (B,3,H,W), src_full (B,3*n_src,H,W): float32, contiguous, on the bound device -- the CALLER vouches for that
        (links.SFMLearnerLoss validates once per set of arrays); nothing is checked here but what the library checks itself."""
        self._zero_d_src()
        loss5 = self.loss5 if out is None else out
        fn = lib.sfm_step_fwd_bwd if grad else lib.sfm_step_fwd
        l5 = self._loss5_arg if out is None else C.c_void_p(out.data_ptr())
        idx = self.device.index
        if torch.cuda.current_device() == idx:
            check(fn(tgt_full.data_ptr(), src_full.data_ptr(), self._desc_ref, l5, self._ws_arg, self._ws_bytes, _stream(idx)))
        else:
            with torch.cuda.device(self.device):
                check(fn(tgt_full.data_ptr(), src_full.data_ptr(), self._desc_ref, l5, self._ws_arg, self._ws_bytes, _stream(idx)))
        return loss5

    def forward(self, out=None):
        """`out`: as for forward_backward."""
        loss5 = self.loss5 if out is None else out
        self._launch(lib.sfm_loss_fwd, self._loss5_arg if out is None else _p(out))
        return loss5

    def backward(self, gy=1.0):
        self._zero_d_src()
        self._launch(lib.sfm_loss_bwd, float(gy))
        return self.d_disps, self.d_poses, self.d_masks, self.d_srcs

    def forward_backward(self, out=None, variant=0):
        """`out`: optional (5,) float32 device tensor to receive the five scalars instead of `self.loss5`
        (lets a caller keep a log of the steps of a reporting interval and reduce it across ranks once).
        `variant`: development hook (sfm_loss_variant): 3 = the kernels read their header from the argument struct."""
        self._zero_d_src()
        loss5 = self.loss5 if out is None else out
        if variant:
            check(lib.sfm_loss_variant(int(variant)))
        self._launch(lib.sfm_loss_fwd_bwd, self._loss5_arg if out is None else _p(out))
        return loss5
