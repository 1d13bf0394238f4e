#!/usr/bin/env python3
"""Where the host time of one SFMLearnerLoss.__call__ + backward() goes (cProfile; the reference's regime: B=4, L1 only).
    python tools/link_profile.py [workload=ref_b4] [graph]"""
import cProfile
import importlib
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bench = importlib.import_module("bench")
PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops"); synth = importlib.import_module(PKG + ".synth")
links = importlib.import_module(PKG + ".links"); cs = importlib.import_module(PKG + ".chainer_surface")
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "ref_b4"
graph = "graph" in sys.argv[2:]
r = bench.Runner(torch, np, ops, synth, dev, wl, "hwc", "fused")
model = links.SFMLearnerLoss(dict(seq_len=r.n_src + 1, smooth_reg=r.cfg.get("smooth_reg", 0.0), exp_reg=0.0, ssim_rate=r.cfg.get("ssim_rate", 0.0)),
                             smooth_mode=r.cfg.get("smooth_mode", "second_order"), use_graph=graph)
K, disps, poses = r.common
vd, vp = [cs.Variable(a) for a in disps], [cs.Variable(a) for a in poses]
tgt, src = r.full


def step():
    for v in vd + vp:
        v.cleargrad()
    loss = model(tgt, src, K, None, vd, vp)
    loss.backward()


for _ in range(50):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
