#!/usr/bin/env python3
"""Diagnostics: where finalize_kernel spends its time (needs a -DSFM_FIN_STAMPS build: SFMWARP_LIB=...).
Stamps are 100 MHz ticks; printed relative to the end of the last wave of the main kernel."""
import importlib, sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops"); synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda:0")
TB, TH, TW, TS = [int(v) for v in os.environ.get("SFM_TRACE_SHAPE", "32,128,416,2").split(",")]
d = synth.make_inputs(B=TB, H=TH, W=TW, n_src=TS, n_scales=4, seed=1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cv = lambda a: ops.to_hwc(t(a))
fl = ops.FusedLoss(smooth_reg=0.1, ssim_rate=0.15, smooth_mode="edge_aware").bind([cv(a) for a in d["tgt_pyr"]], [cv(a) for a in d["src_pyr"]], t(d["intrinsics"]),
                                                                              [t(a) for a in d["disps"]], [t(a) for a in d["poses"]], layout="hwc")
for _ in range(5): fl.forward_backward()
buf = torch.zeros((60000, 4), dtype=torch.int64, device=dev)
for rep in range(3):
    buf.zero_()
    ops.lib.sfm_loss_debug_trace(C.c_void_p(buf.data_ptr()))      # (the hook holds for the next launch only)
    fl.forward_backward(); torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(-1)
    items = raw[:160000].reshape(-1, 4)
    nz = (items != 0).any(axis=1)
    t_end = items[nz][:, 1].max()
    t_beg = items[nz][:, 0].min()
    f = raw[200000:200016]
    us = lambda v: (int(v) - int(t_end)) / 100.0 if v else float("nan")
    print("main kernel waves: first start %.2f us before the last wave's end" % ((t_end - t_beg) / 100.0))
    print("  pose block 0: start %+.2f  tiles known %+.2f  rotation built %+.2f  partials folded %+.2f  wave sums %+.2f  d_pose stored %+.2f" % tuple(us(v) for v in f[0:6]))
    print("  loss block  : start %+.2f  partials summed %+.2f  after barrier %+.2f  loss5 stored %+.2f" % tuple(us(v) for v in f[8:12]))
