#!/usr/bin/env python3
"""Does the BASE address of the source pyramid relative to the target pyramid matter?  (256x832: every image is a multiple of 64 KiB
long, so with 2-MiB-aligned torch allocations the texel of pixel (y, x) lies at the same offset modulo 64 KiB in the target and in
every source image.)  The source arrays are re-allocated with a leading pad of `delta` bytes; everything else is untouched.

    python tools/base_offset_probe.py [workload=cfg5_2src]
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
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg5_2src"
B, H, W, n_src, n_scales, cfg, desc = bench.WORKLOADS[wl]
d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=n_scales, seed=1)
ev = bench.HipEvents()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
tgt = [ops.to_hwc(t(a)) for a in d["tgt_pyr"]]
src0 = [ops.to_hwc(t(a)) for a in d["src_pyr"]]
K, disps, poses = t(d["intrinsics"]), [t(a) for a in d["disps"]], [t(a) for a in d["poses"]]
keep = []


def shifted(a, delta):
    if delta == 0:
        return a
    buf = torch.empty((a.numel() * 4 + delta + 256,), dtype=torch.uint8, device=dev)
    keep.append(buf)
    off = (-buf.data_ptr()) % 256 + delta
    v = buf[off:off + a.numel() * 4].view(torch.float32).view(a.shape)
    v.copy_(a)
    return v


def measure(fl, k=25, blocks=8):
    pair = [ev.create(), ev.create()]
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.4:
        for _ in range(50):
            fl.forward_backward()
        torch.cuda.synchronize()
    ts, ks = [], []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for i in range(k):
            if i == k // 2:
                ops.lib.sfm_loss_profile_events(pair[0], pair[1])
            fl.forward_backward()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / k * 1e6)
        ks.append(ev.elapsed_ms(pair[0], pair[1]) * 1e3)
    return float(np.median(ts)), float(np.median(ks))


for rep in range(2):
    for delta in (0, 256, 4096 + 256, 32768 + 256, 16384 + 4096 + 256, 0):
        src = [shifted(a, delta * (s + 1)) for s, a in enumerate(src0)]
        fl = ops.FusedLoss(**cfg).bind(tgt, src, K, disps, poses, layout="hwc")
        keep.append(fl)
        step, kern = measure(fl)
        print("%s rep %d: source pyramids shifted by %6d B x (scale + 1): step %.2f us, main kernel %.2f us  (src0 - tgt0 mod 64 KiB: %d)" % (
            wl, rep, delta, step, kern, (src[0].data_ptr() - tgt[0].data_ptr()) % 65536), flush=True)
