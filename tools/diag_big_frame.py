import importlib, os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import sfm_oracle as O
import test_loss_gpu as T
ops = importlib.import_module("sfm-learner-chainer_amd.ops")
synth = importlib.import_module("sfm-learner-chainer_amd.synth")
dev = torch.device("cuda:0")
d = synth.make_inputs(B=1, H=1024, W=1408, n_src=1, n_scales=2, seed=4)
cfg = dict(smooth_reg=0.1, ssim_rate=0.15)
ref = T._oracle(d, cfg)
r64 = O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], None, backward=True, keep_warped=True, dtype=np.float64, **cfg)
fl = T._bind(ops, dev, d, cfg, layout="planar", want_warped=True)
fl.forward_backward()
width = T.cell_width_from_position_bound(d)
for s in range(2):
    g = fl.d_disps[s].cpu().numpy(); w = ref["d_disps"][s]
    knife = T.knife_mask(ref, s, cell_thr=width(s))[0][:, None]
    err = np.abs(g - w) / np.abs(w).max()
    bad = (err > 2e-3) & ~knife
    print("scale", s, "bad", bad.sum())
    for (b, c, y, x) in np.argwhere(bad)[:10]:
        print("  px", y, x, "err %.3e"%err[b,c,y,x], "got %.4e want %.4e w64 %.4e"%(g[b,c,y,x], w[b,c,y,x], r64["d_disps"][s][b,c,y,x]),
              "margin %.2e cell %.2e abs %.2e clip %.2e"%(ref["margin"][s][b,0,y,x], ref["cell_margin"][s][b,0,y,x], ref["abs_margin"][s][b,0,y,x], ref["clip_margin"][s][b,0,y,x]),
              "width %.2e"%width(s)[b,0,y,x], "UV", ref["uv"][s][b,0,:,y,x], "uv64", r64["uv"][s][b,0,:,y,x],
              "warp err", np.abs(fl.warped[s].cpu().numpy()[b,0,:,y,x]-ref["warped"][s][b,0,:,y,x]).max())
        # neighbourhood: any knife pixels in 5x5?
        ys, xs = slice(max(y-2,0), y+3), slice(max(x-2,0), x+3)
        print("     min margins in 5x5: flip %.2e cell %.2e abs %.2e clip %.2e"%(ref["margin"][s][b,0,ys,xs].min(), ref["cell_margin"][s][b,0,ys,xs].min(), ref["abs_margin"][s][b,0,ys,xs].min(), ref["clip_margin"][s][b,0,ys,xs].min()))
