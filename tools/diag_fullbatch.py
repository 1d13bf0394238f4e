#!/usr/bin/env python3
"""Diagnostics: the worst d_disp elements of a full-batch launch vs the fp32 / fp64 oracle, with the knife-edge margins around them."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_loss_gpu as T
from oracle import sfm_oracle as O
ops = importlib.import_module("sfm-learner-chainer_amd.ops"); synth = importlib.import_module("sfm-learner-chainer_amd.synth")
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "ssim_smooth"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = T.CONFIGS[cfg_name]
dev = torch.device("cuda:0")
# SFM_DIAG_SHAPE="H,W,n_src,n_scales,seed" overrides the BASELINE shape
DH, DW, DN, DS, DSEED = [int(v) for v in os.environ.get("SFM_DIAG_SHAPE", "128,416,2,4,1").split(",")]
d = synth.make_inputs(B=B, H=DH, W=DW, n_src=DN, n_scales=DS, seed=DSEED)
ref = T._oracle(d, cfg)
fl = T._bind(ops, dev, d, cfg, layout="hwc")
fl.forward_backward()
for s in range(DS):
    g = fl.d_disps[s].cpu().numpy().astype(np.float64); w = ref["d_disps"][s].astype(np.float64)
    knife = T.knife_mask(ref, s, cell_thr=float(os.environ.get("SFM_DIAG_CELL_THR", "1e-4")))[0][:, None]
    err = np.abs(g - w) * ~knife / np.abs(w).max()
    idx = np.argsort(err.ravel())[::-1][:3]
    for j in idx:
        b, _, y, x = np.unravel_index(j, err.shape)
        if err[b, 0, y, x] < 5e-4: continue
        print("scale %d sample %d (y=%d, x=%d): kernel %.6e oracle32 %.6e rel err %.4f  max|ref| %.3e" % (s, b, y, x, g[b, 0, y, x], w[b, 0, y, x], err[b, 0, y, x], np.abs(w).max()))
        sl = (slice(max(y - 3, 0), y + 4), slice(max(x - 3, 0), x + 4))
        for i in range(DN):
            print("  src %d margin min %.2e clip min %.2e cell min %.2e abs min %.2e | U,V at pixel %.4f %.4f  disp %.5f" % (
                i, ref["margin"][s][b, i][sl].min(), ref["clip_margin"][s][b, i][sl].min(), ref["cell_margin"][s][b, i][sl].min(),
                ref["abs_margin"][s][b, i][sl].min(), ref["uv"][s][b, i, 0, y, x], ref["uv"][s][b, i, 1, y, x], d["disps"][s][b, 0, y, x]))
            print("  src %d warped at px" % i, ref["warped"][s][b, i, :, y, x], "tgt", d["tgt_pyr"][s][b, :, y, x])
        # one-sample fp64
        one = lambda a: a[b:b + 1]
        r64 = O.sfm_loss([one(a) for a in d["tgt_pyr"]], [one(a) for a in d["src_pyr"]], one(d["intrinsics"]), [one(a) for a in d["disps"]],
                         [one(a) for a in d["poses"]], None, backward=True, dtype=np.float64, norm_batch=B, **cfg)
        print("  fp64 oracle: %.6e" % r64["d_disps"][s][0, 0, y, x])
        print("  kernel 5x5 d_disp:\n", np.array2string(g[b, 0, y - 2:y + 3, x - 2:x + 3], precision=3))
        print("  oracle 5x5 d_disp:\n", np.array2string(w[b, 0, y - 2:y + 3, x - 2:x + 3], precision=3))
