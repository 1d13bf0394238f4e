#!/usr/bin/env python3
"""Worst warped pixels of the fused kernel (SfmLossDesc.warped) against the fp32 / fp64 oracle, with what the sample looks like there:
position, local source gradient, the displacement the error implies, depth and z.   python tools/diag_warped.py [motion] [B]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import sfm_oracle as O
import test_loss_gpu as T
ops = importlib.import_module("sfm-learner-chainer_amd.ops")
synth = importlib.import_module("sfm-learner-chainer_amd.synth")
motion = sys.argv[1] if len(sys.argv) > 1 else "small"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W, n_src, S = [int(v) for v in os.environ.get("SFM_DIAG_SHAPE", "128,416,2,4").split(",")]
dev = torch.device("cuda:0")
cfg = T.CONFIGS["edge_aware"]
d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=S, seed=1) if motion == "small" else T.make_motion_inputs(synth, motion, B=B, H=H, W=W, n_src=n_src, n_scales=S, seed=21)
ref = T._oracle(d, cfg)
r64 = O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], None, keep_warped=True, dtype=np.float64, **cfg)
fl = T._bind(ops, dev, d, cfg, layout="hwc", want_warped=True)
fl.forward_backward()
for s in range(S):
    g = fl.warped[s].cpu().numpy()
    w, w64 = ref["warped"][s], r64["warped"][s]
    e32, e64, own = np.abs(g - w).max(axis=2), np.abs(g - w64).max(axis=2), np.abs(w - w64).max(axis=2)
    mism = (g == 0).all(axis=2) != (w == 0).all(axis=2)
    e32[mism] = 0; e64[mism] = 0
    print("scale %d: max err vs fp32 oracle %.2e, vs fp64 %.2e; fp32 oracle's own %.2e; >1e-4: %d (fp32), %d (fp64), oracle's own %d" % (
        s, e32.max(), e64.max(), own.max(), (e32 > 1e-4).sum(), (e64 > 1e-4).sum(), (own > 1e-4).sum()))
    if s: continue
    src = d["src_pyr"][s]
    uv, uv64 = ref["uv"][s], r64["uv"][s]
    order = np.argsort(e64.ravel())[::-1][:12]
    for k in order:
        b, i, y, x = np.unravel_index(k, e64.shape)
        U, V = uv[b, i, 0, y, x], uv[b, i, 1, y, x]
        u0, v0 = int(np.floor(U)), int(np.floor(V))
        im = src[b, 3 * i:3 * i + 3]
        gx = np.abs(im[:, v0:v0 + 2, u0 + 1] - im[:, v0:v0 + 2, u0]).max()
        gy = np.abs(im[:, v0 + 1, u0:u0 + 2] - im[:, v0, u0:u0 + 2]).max()
        disp = d["disps"][s][b, 0, y, x]
        print("  (b%d i%d y%d x%d) err32 %.2e err64 %.2e own %.2e | U %.4f V %.4f (fp64 %.5f %.5f: dU %.1e dV %.1e) | |dI/du| %.3f |dI/dv| %.3f -> implied shift %.1e px | disp %.3f D %.3f" % (
            b, i, y, x, e32[b, i, y, x], e64[b, i, y, x], own[b, i, y, x], U, V, uv64[b, i, 0, y, x], uv64[b, i, 1, y, x], U - uv64[b, i, 0, y, x], V - uv64[b, i, 1, y, x],
            gx, gy, e64[b, i, y, x] / max(gx, gy, 1e-9), disp, 1 / disp))
