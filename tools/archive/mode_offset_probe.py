#!/usr/bin/env python3
"""Does the level of a cfg3 step (tools/mode_batch_probe.py: it changes from one set of arrays to the next inside a process) follow the
RELATIVE placement of the output arrays?  One process, inputs fixed: d_disp of every scale is moved through one big buffer, its
distance to disp[s] (same sample, row, column = same offset inside the array) modulo 2 MiB taking chosen values.

    python tools/mode_offset_probe.py
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


def take(runner, k=20, blocks=8):
    pair = [ev.create(), ev.create()]
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.12:
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


R = bench.Runner(torch, np, ops, synth, dev, "cfg3_edge", "hwc", "fused")
fl = R.fl
tgt, src, K, disps, poses, _ = fl._keep
M2 = 1 << 21
step, kern = take(R)
rel = lambda a, b: (a.data_ptr() - b.data_ptr()) % M2
print("as allocated: step %.2f kernel %.2f | (d_disp0 - disp0) mod 2M = %dK, (src0 - tgt0) mod 2M = %dK, (disp0 - tgt0) mod 2M = %dK, (d_disp0 - tgt0) = %dK" % (
    step, kern, rel(fl.d_disps[0], disps[0]) >> 10, rel(src[0], tgt[0]) >> 10, rel(disps[0], tgt[0]) >> 10, rel(fl.d_disps[0], tgt[0]) >> 10), flush=True)
big = torch.empty((96 << 20,), dtype=torch.uint8, device=dev)
base = (big.data_ptr() + M2 - 1) // M2 * M2
sizes = [t.numel() * 4 for t in fl.d_disps]
for off_k in (0, 1, 4, 16, 64, 128, 256, 512, 768, 1024, 1536, 1, 0, 640, 96, 1000):
    # d_disp[s] at: base + s * 16 MiB + (disp[s] mod 2M) + off   => (d_disp[s] - disp[s]) mod 2M = off for every scale
    new = []
    for s, t in enumerate(fl.d_disps):
        want = (disps[s].data_ptr() + (off_k << 10)) % M2
        start = base + s * (16 << 20) + want - big.data_ptr()
        view = big[start:start + sizes[s]].view(torch.float32).view(t.shape)
        new.append(view)
        fl.desc.d_disp[s] = view.data_ptr()
    fl.d_disps = new
    step, kern = take(R)
    print("(d_disp - disp) mod 2M = %5dK: step %.2f kernel %.2f" % (off_k, step, kern), flush=True)
