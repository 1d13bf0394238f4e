#!/usr/bin/env python3
"""The slow level of a cfg3 step hits only some workloads of a process (profiles/r05_process_modes.txt: headline and cfg3 slow, every
other key of the same bench line -- the same arrays with larger poses or a smooth disparity, the same bytes as 8 x 256x832 -- unchanged).
Which?  One process: cfg3 as written at B = 32 / 31 / 33 / 24 / 16, the second-order form, a second seed, the planar layout.

    python tools/mode_batch_probe.py
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bench = importlib.import_module("bench")
PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops")
synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", 0)
ev = bench.HipEvents()


def take(runner, k=20, blocks=10):
    pair = [ev.create(), ev.create()]
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.15:
        for _ in range(50):
            runner.step()
        torch.cuda.synchronize()
    ts, ks = [], []
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            runner.step(evs=pair if i == k // 2 else None)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / k * 1e6)
        ks.append(ev.elapsed_ms(pair[0], pair[1]) * 1e3)
    return float(np.median(ts)), float(np.median(ks))


out = []
for label, kw in (("B=32", dict()), ("B=31", dict(batch=31)), ("B=33", dict(batch=33)), ("B=24", dict(batch=24)), ("B=16", dict(batch=16)),
                  ("B=32 again", dict()), ("B=32 seed 2", dict(seed=2)), ("cfg3 (2nd order)", dict(workload="cfg3")), ("planar", dict(layout="planar"))):
    wl = kw.pop("workload", "cfg3_edge")
    layout = kw.pop("layout", "hwc")
    R = bench.Runner(torch, np, ops, synth, dev, wl, layout, "fused", **kw)
    step, kern = take(R)
    px = R.warped_px
    out.append("%s: %.2f / %.2f us (%.1f ns per kpx)" % (label, step, kern, kern * 1e3 / (px / 1e3)))
    del R
print(" | ".join(out), flush=True)
