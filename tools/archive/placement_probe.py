#!/usr/bin/env python3
"""Does the step time depend on WHERE the arrays lie relative to each other?  (round 5: two processes on one box ran the same build at
59.8 and 56.9 us per step, each rock-steady over 4 s.)  All device arrays of a cfg3 step are carved out of ONE arena at controlled
offsets: array k starts at the next multiple of `align` plus k * `stagger` bytes.  Prints the step and main-kernel time per setting.

    python tools/placement_probe.py [workload]
"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bench = importlib.import_module("bench")
PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops")
synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_edge"
B, H, W, n_src, n_scales, cfg, desc = bench.WORKLOADS[wl]
d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=n_scales, seed=1)
ev = bench.HipEvents()
host = lambda a: torch.from_numpy(np.ascontiguousarray(a))
hwc = lambda a: host(a).reshape(a.shape[0], a.shape[1] // 3, 3, a.shape[2], a.shape[3]).permute(0, 1, 3, 4, 2).contiguous()
arrays = [("tgt%d" % s, hwc(a)) for s, a in enumerate(d["tgt_pyr"])] + [("src%d" % s, hwc(a)) for s, a in enumerate(d["src_pyr"])] + \
         [("disp%d" % s, host(a)) for s, a in enumerate(d["disps"])] + [("K", host(d["intrinsics"]))] + [("pose%d" % i, host(a)) for i, a in enumerate(d["poses"])]
total = sum(a.numel() * 4 for _, a in arrays)


class Timer:
    def __init__(self, fl):
        self.fl = fl

    def run(self, k=25, blocks=8):
        fl = self.fl
        pairs = [ev.create(), ev.create()]
        import time
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.4:      # (warm: the first steps after an idle second run 10 % slower for some tens of milliseconds)
            for _ in range(50):
                fl.forward_backward()
            torch.cuda.synchronize()
        ts, ks = [], []
        for _ in range(blocks):
            t0 = time.perf_counter()
            for i in range(k):
                if i == k // 2:
                    ops.lib.sfm_loss_profile_events(pairs[0], pairs[1])
                fl.forward_backward()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / k * 1e6)
            ks.append(ev.elapsed_ms(pairs[0], pairs[1]) * 1e3)
        return float(np.median(ts)), float(np.median(ks))


keep = []      # (everything stays allocated: every take gets fresh addresses, outputs and workspace included)


def arena_take(align, stagger, only=None):
    """`only`: names (prefixes) of the arrays that go into the staggered arena; the others become separate torch allocations."""
    arena = torch.empty((total + (len(arrays) + 2) * (align + (1 << 20)) + len(arrays) * stagger * len(arrays),), dtype=torch.uint8, device=dev)
    keep.append(arena)
    base = arena.data_ptr()
    off = (-base) % (1 << 21)
    views = {}
    for k, (name, a) in enumerate(arrays):
        if only is not None and not name.startswith(only):
            views[name] = a.to(dev)
            continue
        off = (off + align - 1) // align * align + k * stagger
        nbytes = a.numel() * 4
        v = arena[off:off + nbytes].view(torch.float32).view(a.shape)
        v.copy_(a)
        views[name] = v
        off += nbytes
    fl = ops.FusedLoss(**cfg).bind([views["tgt%d" % s] for s in range(n_scales)], [views["src%d" % s] for s in range(n_scales)], views["K"],
                                   [views["disp%d" % s] for s in range(n_scales)], [views["pose%d" % i] for i in range(n_src)], layout="hwc")
    keep.append(fl)
    return fl, "arena: %s at multiples of %d B + k x %d B" % ("all arrays" if only is None else "/".join(only) + " only", align, stagger)


def torch_take():
    R = bench.Runner(torch, np, ops, synth, dev, wl, "hwc", "fused")
    keep.append(R)
    return R.fl, "separate torch allocations (bench.py)"


TAKES = [torch_take, lambda: arena_take(1 << 21, 0), lambda: arena_take(256, 0), lambda: arena_take(1 << 21, 65536 + 4096 + 256), lambda: arena_take(1 << 21, (1 << 18) + 4096 + 256)]
for rep in range(3):
    for take in (TAKES if rep != 1 else TAKES[::-1]):
        fl, what = take()
        step, kern = Timer(fl).run()
        ins = [t.data_ptr() for t in fl._keep[0][:1] + fl._keep[1][:1] + fl._keep[3][:1]] + [fl.d_disps[0].data_ptr(), fl.ws.data_ptr()]
        print("%s take %d: %-78s step %.2f us, main kernel %.2f us | tgt0 src0 disp0 d_disp0 ws at (MiB) %s" % (
            wl, rep, what, step, kern, " ".join("%.1f" % ((p_ - ins[0]) / 2.0 ** 20) for p_ in ins)), flush=True)
