#!/usr/bin/env python3
"""Host-side: the work decomposition the library chooses (sfm_loss_plan_info).  usage: tools/show_plan.py B H W n_src n_scales"""
import ctypes as C
import importlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("sfm-learner-chainer_amd._lib")


def plan(B, H, W, n_src, S, ssim=0.15, smooth=0.1, grad=1, loss=1, smooth_mode=1, hwc=True):
    d = _lib.SfmLossDesc()
    d.B, d.norm_B, d.n_src, d.n_scales, d.ssim_rate, d.smooth_reg, d.smooth_mode = B, max(B, 1), n_src, S, ssim, smooth, smooth_mode
    d.intrinsics = 1
    d.image_layout = _lib.SFM_LAYOUT_HWC if hwc else _lib.SFM_LAYOUT_PLANAR      # (hwc: what the link and bench.py bind)
    for s in range(S):
        d.H[s], d.W[s] = H >> s, W >> s
        d.tgt[s] = d.src[s] = d.disp[s] = d.d_disp[s] = 1      # never dereferenced: nothing is launched
    for i in range(n_src):
        d.pose[i] = d.d_pose[i] = 1
    out = (C.c_int * (1 + 4 * S))()
    rc = _lib.lib.sfm_loss_plan_info(C.byref(d), grad, loss, out, len(out))
    if rc != 0:
        raise RuntimeError(_lib.last_error())
    return out[0], [dict(zip(("strips", "chunks", "rows", "tiles"), out[1 + 4 * s:5 + 4 * s])) for s in range(S)]


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:6]] or [32, 128, 416, 2, 4]
    items, scales = plan(*a)
    print("items", items)
    for s, sc in enumerate(scales):
        print("scale", s, sc)
