import importlib, os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import sfm_oracle as O
import test_loss_gpu as T
ops = importlib.import_module("sfm-learner-chainer_amd.ops")
synth = importlib.import_module("sfm-learner-chainer_amd.synth")
dev = torch.device("cuda:0")
cfg = T.CONFIGS["ssim_smooth"]
d = synth.make_inputs(B=2, H=32, W=104, n_src=2, n_scales=2, seed=3)
for k in ("tgt_pyr", "src_pyr"):
    d[k] = [(a * 127.5 + 127.5).astype(np.float32) for a in d[k]]
for layout in ("planar", "hwc"):
    fl = T._bind(ops, dev, d, cfg, layout=layout)
    l = fl.forward_backward().cpu().numpy()
    r32 = O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], None, backward=True, **cfg)
    r64 = O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], None, backward=True, dtype=np.float64, **cfg)
    print(layout, "0..255: loss5", l, "oracle32", [r32[k] for k in T.KEYS], "oracle64", [r64[k] for k in T.KEYS])
    g = fl.d_disps[0].cpu().numpy()
    print("   d_disp finite", np.isfinite(g).all(), "rel L2 vs 64: kernel %.2e oracle32 %.2e" % (T.rel_l2(g, r64["d_disps"][0]), T.rel_l2(r32["d_disps"][0], r64["d_disps"][0])))
d = synth.make_inputs(B=2, H=32, W=104, n_src=2, n_scales=2, seed=3)
d["src_pyr"][0][0, 1, 10, 40] = np.nan
for layout in ("planar", "hwc"):
    fl = T._bind(ops, dev, d, cfg, layout=layout)
    l = fl.forward_backward().cpu().numpy()
    r32 = O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], None, backward=True, **cfg)
    print(layout, "NaN pixel: loss5", l, "oracle32", [r32[k] for k in T.KEYS], "d_disp finite share", np.isfinite(fl.d_disps[0].cpu().numpy()).mean(), "d_pose", fl.d_poses[0].cpu().numpy()[0])
