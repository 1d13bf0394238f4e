#!/usr/bin/env python3
"""The 57.5 / 60.9 us modes of a cfg3 step follow the ALLOCATION of its arrays (tools/queue_mode_probe.py: a process whose first set of
arrays runs fast gets a slow one on re-allocation with probability ~1/3).  Which array, and what about its address?  One process:
N complete re-allocations (every earlier set stays alive), each timed, with the device addresses of its arrays.

    python tools/alloc_mode_probe.py [N]
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10


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


def addr(t):
    p = t.data_ptr()
    return "%x(+%4dK of 2M)" % (p >> 21, (p & ((1 << 21) - 1)) >> 10)


keep = []
for i in range(N):
    R = bench.Runner(torch, np, ops, synth, dev, "cfg3_edge", "hwc", "fused")
    keep.append(R)
    fl = R.fl
    tgt, src, K, disps, poses, _ = fl._keep
    step, kern = take(R)
    print("set %2d  step %.2f kernel %.2f  %s | tgt0 %s src0 %s src1 %s disp0 %s d_disp0 %s ws %s d_pose0 %s" % (
        i, step, kern, "SLOW" if kern > 55.3 else "fast", addr(tgt[0]), addr(src[0]), addr(src[1]), addr(disps[0]), addr(fl.d_disps[0]), addr(fl.ws),
        addr(fl.d_poses[0])), flush=True)
# the slow sets again, and the fast ones: is the mode a property of the set, for good?
for i in (0, N // 2, N - 1):
    step, kern = take(keep[i])
    print("set %2d again  step %.2f kernel %.2f" % (i, step, kern), flush=True)
