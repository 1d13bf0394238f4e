#!/usr/bin/env python3
"""Host-side cost of one SFMLearnerLoss.__call__ + backward() (pyramid + loss launches): a small problem (the GPU is never the limit)
and BASELINE cfg3; enqueue time per step vs time including the final synchronisation."""
import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "sfm-learner-chainer_amd"
links = importlib.import_module(PKG + ".links"); cs = importlib.import_module(PKG + ".chainer_surface"); synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda:0")
for (B, H, W) in ((1, 64, 96), (4, 128, 416), (32, 128, 416)):
    d = synth.make_inputs(B=B, H=H, W=W, n_src=2, n_scales=4, seed=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tgt = t(d["tgt_pyr"][0]); src = t(d["src_pyr"][0]).reshape(B, 2, 3, H, W)
    K = t(d["intrinsics"])
    vd = [cs.Variable(t(a)) for a in d["disps"]]; vp = [cs.Variable(t(a)) for a in d["poses"]]
    model = links.SFMLearnerLoss(dict(seq_len=3, smooth_reg=0.1, exp_reg=0.0, ssim_rate=0.15), smooth_mode="edge_aware")
    def step():
        for v in vd + vp: v.cleargrad()
        loss = model(tgt, src, K, None, vd, vp)
        loss.backward()
    for _ in range(20): step()
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(200): step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
    print("B=%d %dx%d, the same arrays every call (fast path): host enqueue per step %.1f us, with final sync %.1f us" % (
        B, H, W, np.median([r[0] for r in res]), np.median([r[1] for r in res])))
    # ... and as a training loop presents it: NEW arrays for the network outputs every step (same shapes): validation + re-binding
    pool = [([cs.Variable(v.data.clone()) for v in vd], [cs.Variable(v.data.clone()) for v in vp]) for _ in range(4)]
    def step_new(k):
        d_, p_ = pool[k % 4]
        for v in d_ + p_: v.cleargrad()
        loss = model(tgt, src, K, None, d_, p_)
        loss.backward()
    for k in range(20): step_new(k)
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        t0 = time.perf_counter()
        for k in range(200): step_new(k)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
    print("B=%d %dx%d, other arrays every call (full path):     host enqueue per step %.1f us, with final sync %.1f us" % (
        B, H, W, np.median([r[0] for r in res]), np.median([r[1] for r in res])))
