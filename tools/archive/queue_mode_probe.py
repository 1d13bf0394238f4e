#!/usr/bin/env python3
"""Is the per-process 'mode' of the step time (57.5 vs 60.5 us at cfg3 as written, rock-steady inside a process, random between
processes: tools/kernarg_ab.sh) a property of the HIP stream (hardware queue) the launches go through?  One process: the same bound
step timed on torch's default stream, then on freshly created streams, then with every array re-allocated.

    python tools/queue_mode_probe.py            (prints one line per take)
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


def take(runner, label, k=20, blocks=12):
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
    print("%-44s step %.2f us   main kernel %.2f us" % (label, float(np.median(ts)), float(np.median(ks))), flush=True)


R = bench.Runner(torch, np, ops, synth, dev, "cfg3_edge", "hwc", "fused")
take(R, "default stream")
streams = [torch.cuda.Stream(device=dev) for _ in range(6)]
for i, s in enumerate(streams):
    with torch.cuda.stream(s):
        take(R, "new stream %d (0x%x)" % (i, s.cuda_stream))
take(R, "default stream, after the others")
keep = [R]
for i in range(1):
    R2 = bench.Runner(torch, np, ops, synth, dev, "cfg3_edge", "hwc", "fused")
    keep.append(R2)
    take(R2, "default stream, every array re-allocated (%d)" % i)
