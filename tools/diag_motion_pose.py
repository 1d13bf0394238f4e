#!/usr/bin/env python3
"""Diagnostics (round 6): d_pose of both projections on a MOTION case of tests/test_loss_gpu.py, per element, against the fp32 and the
fp64 oracle, and the pixels that carry the difference.    python tools/diag_motion_pose.py [motion] [cfg] [seed]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_loss_gpu as T
from oracle import sfm_oracle as O
ops = importlib.import_module("sfm-learner-chainer_amd.ops"); synth = importlib.import_module("sfm-learner-chainer_amd.synth")
motion = sys.argv[1] if len(sys.argv) > 1 else "behind"
cfg_name = sys.argv[2] if len(sys.argv) > 2 else "edge_aware"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 21
cfg = T.CONFIGS[cfg_name]
dev = torch.device("cuda:0")
d = T.make_motion_inputs(synth, motion, B=4, H=128, W=416, n_src=2, n_scales=4, seed=seed, with_masks=True)
ref = T._oracle(d, cfg)
ref64 = O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], d["masks"], backward=True, keep_warped=True, dtype=np.float64, **cfg)
np.set_printoptions(linewidth=200, precision=3)
for proj in ("fast", "reference_order"):
    fl = T._bind(ops, dev, d, cfg, layout="hwc", want_warped=True, projection=proj)
    fl.forward_backward()
    print("==", proj)
    for i in range(2):
        g = fl.d_poses[i].cpu().numpy().astype(np.float64); w = ref["d_poses"][i].astype(np.float64); w64 = ref64["d_poses"][i]
        m = np.abs(w64).max()
        print(" src %d (max %.3e) kernel - fp32 oracle, of the max:\n%s\n   kernel - fp64:\n%s\n   fp32 oracle - fp64:\n%s" % (i, m, (g - w) / m, (g - w64) / m, (w - w64) / m))
    for s in range(4):
        gw = fl.warped[s].cpu().numpy().astype(np.float64)
        for i in range(2):
            d32 = np.abs(gw[:, i] - ref["warped"][s][:, i]).max(axis=1)       # (B,h,w)
            flips = ((gw[:, i] == 0).all(axis=1) != (ref["warped"][s][:, i] == 0).all(axis=1))
            j = np.argmax(d32 * ~flips)
            b, y, x = np.unravel_index(j, d32.shape)
            print(" scale %d src %d: flips %d; worst warped diff %.2e at sample %d (%d,%d): z oracle %.4e U,V %.3f %.3f margin %.2e" % (
                s, i, int(flips.sum()), (d32 * ~flips).max(), b, y, x, float(ref["z"][s][b, i, y, x]) if "z" in ref else float("nan"),
                ref["uv"][s][b, i, 0, y, x], ref["uv"][s][b, i, 1, y, x], ref["margin"][s][b, i, y, x]))
    # d_disp: the largest differences to the fp32 oracle anywhere (knife pixels included), with the sample they sit in
    for s in range(4):
        g = fl.d_disps[s].cpu().numpy().astype(np.float64); w = ref["d_disps"][s].astype(np.float64); w64 = ref64["d_disps"][s]
        e = np.abs(g - w)[:, 0]
        idx = np.argsort(e.ravel())[::-1][:3]
        for j in idx:
            b, y, x = np.unravel_index(j, e.shape)
            print(" scale %d d_disp diff %.3e (max %.3e) at sample %d (%d,%d): kernel %.4e o32 %.4e o64 %.4e | margins src0/1 flip %.1e %.1e cell %.1e %.1e abs %.1e %.1e" % (
                s, e[b, y, x], np.abs(w).max(), b, y, x, g[b, 0, y, x], w[b, 0, y, x], w64[b, 0, y, x],
                ref["margin"][s][b, 0, max(y-2,0):y+3, max(x-2,0):x+3].min(), ref["margin"][s][b, 1, max(y-2,0):y+3, max(x-2,0):x+3].min(),
                ref["cell_margin"][s][b, 0, y, x], ref["cell_margin"][s][b, 1, y, x], ref["abs_margin"][s][b, 0, y, x], ref["abs_margin"][s][b, 1, y, x]))
