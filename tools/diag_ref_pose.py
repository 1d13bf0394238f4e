#!/usr/bin/env python3
"""Diagnostics (round 6): where d_pose / d_disp of the REFERENCE_ORDER projection differ from the fp32 oracle at cfg5 (4 sources).
    python tools/diag_ref_pose.py [n_src] [seam]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_loss_gpu as T
from oracle import sfm_oracle as O
ops = importlib.import_module("sfm-learner-chainer_amd.ops"); synth = importlib.import_module("sfm-learner-chainer_amd.synth")
n_src = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seam = sys.argv[2] if len(sys.argv) > 2 else "shift"
cfg = T.CONFIGS["ssim_smooth"]
dev = torch.device("cuda:0")
B, H, W = 8, 256, 832
d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=4, seed=1, seam=seam)
ref = T._oracle(d, cfg)
ref64 = O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], None, backward=True, dtype=np.float64, **cfg)
for proj in ("fast", "reference_order"):
    fl = T._bind(ops, dev, d, cfg, layout="hwc", want_warped=True, projection=proj)
    fl.forward_backward()
    print("==", proj)
    for i in range(n_src):
        g = fl.d_poses[i].cpu().numpy().astype(np.float64)
        w, w64 = ref["d_poses"][i].astype(np.float64), ref64["d_poses"][i]
        m = np.abs(w).max()
        print(" src %d: max|d_pose| %.3e; per-sample worst element vs fp32: %s" % (i, m, np.array2string(np.abs(g - w).max(axis=1) / m, precision=2)))
        print("         vs fp64: %s ; oracle32 vs fp64: %s" % (np.array2string(np.abs(g - w64).max(axis=1) / m, precision=2), np.array2string(np.abs(w - w64).max(axis=1) / m, precision=2)))
    # positions: how many warped pixels differ at all, per source, scale 0
    for s in range(4):
        gw = fl.warped[s].cpu().numpy()
        diff = np.abs(gw.astype(np.float64) - ref["warped"][s]).max(axis=2)       # (B,n,h,w)
        print(" scale %d: warped max diff per source %s" % (s, np.array2string(diff.reshape(B, n_src, -1).max(axis=(0, 2)), precision=2)))
        g = fl.d_disps[s].cpu().numpy().astype(np.float64); w = ref["d_disps"][s].astype(np.float64)
        knife = T.knife_mask(ref, s)[0][:, None]
        err = np.abs(g - w) * ~knife / np.abs(w).max()
        j = np.argmax(err)
        b, _, y, x = np.unravel_index(j, err.shape)
        print("   d_disp worst outside the knife mask %.2e at sample %d (y=%d, x=%d): kernel %.4e oracle32 %.4e oracle64 %.4e" % (
            err.ravel()[j], b, y, x, g[b, 0, y, x], w[b, 0, y, x], ref64["d_disps"][s][b, 0, y, x]))
        for i in range(n_src):
            print("     src %d: margin %.2e clip %.2e cell %.2e abs %.2e  warped diff at px %.2e  U,V %.4f %.4f" % (
                i, ref["margin"][s][b, i, max(y-2,0):y+3, max(x-2,0):x+3].min(), ref["clip_margin"][s][b, i, max(y-1,0):y+2, max(x-1,0):x+2].min(), ref["cell_margin"][s][b, i, y, x],
                ref["abs_margin"][s][b, i, y, x], diff[b, i, y, x], ref["uv"][s][b, i, 0, y, x], ref["uv"][s][b, i, 1, y, x]))
