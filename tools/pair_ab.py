#!/usr/bin/env python3
"""In-process A/B of the two forms of the SSIM gradient launch (round 6): one source per pass at three waves per SIMD
(sfm_loss_variant 4) against two sources per pass at two waves per SIMD (variant 5, loss_kernel_pair).  Interleaved rounds in ONE
process: main kernel by HIP events on the dispatch, whole step by wall clock over blocks.    python tools/pair_ab.py [--rounds 6]"""
import argparse, importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")
ops = importlib.import_module("sfm-learner-chainer_amd.ops"); synth = importlib.import_module("sfm-learner-chainer_amd.synth")
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--workloads", default="cfg3_edge,cfg3,cfg3_smooth_disp,cfg3_large_motion,cfg5_2src,cfg5,b16,b8,b4")
args = ap.parse_args()
dev = torch.device("cuda", 0)
ev = bench.HipEvents()
e0, e1 = ev.create(), ev.create()
lines = []
def say(s):
    print(s, flush=True); lines.append(s)
for wl in args.workloads.split(","):
    batch = 0
    name = wl
    if wl.startswith("b") and wl[1:].isdigit():
        name, batch = "cfg3_edge", int(wl[1:])
    R = bench.Runner(torch, np, ops, synth, dev, name, "hwc", "fused", batch)
    fl = R.fl
    def step(v):
        fl.forward_backward(variant=v)
    res = {4: {"k": [], "s": []}, 5: {"k": [], "s": []}}
    for v in (4, 5):
        for _ in range(10):
            step(v)
    bench.warm_inputs(R)
    for rnd in range(args.rounds):
        for v in ((4, 5) if rnd % 2 == 0 else (5, 4)):
            for _ in range(5):
                step(v)
            torch.cuda.synchronize()
            K = 50
            t0 = time.perf_counter()
            for _ in range(K):
                step(v)
            torch.cuda.synchronize()
            res[v]["s"].append((time.perf_counter() - t0) / K * 1e6)
            for _ in range(10):
                ops.lib.sfm_loss_profile_events(e0, e1)
                step(v)
                torch.cuda.synchronize()
                res[v]["k"].append(ev.elapsed_ms(e0, e1) * 1e3)
    m = lambda a: float(np.median(a))
    k4, k5, s4, s5 = m(res[4]["k"]), m(res[5]["k"]), m(res[4]["s"]), m(res[5]["s"])
    px = R.warped_px
    say("%-18s B=%-2d %dx%d %d src | main kernel: one source per pass %7.2f us, two %7.2f us (%+.1f %%) | step %7.2f -> %7.2f us (%+.1f %%) | kernel frac of 8 TB/s %.3f -> %.3f, step %.3f -> %.3f" % (
        wl, R.B, R.H, R.W, R.n_src, k4, k5, 100 * (k5 / k4 - 1), s4, s5, 100 * (s5 / s4 - 1), 60 * px / k4 / 8e6, 60 * px / k5 / 8e6, 60 * px / s4 / 8e6, 60 * px / s5 / 8e6))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "pair_ab.txt"), "w").write("\n".join(lines) + "\n")
