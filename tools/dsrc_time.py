#!/usr/bin/env python3
"""Timing of the fused step with the optional dL/d(src) output bound (development; SFMWARP_LIB selects the build).

    python tools/dsrc_time.py [workload ...] [--only]      (--only: skip the run without d_src)"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bench = importlib.import_module("bench")
PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops"); synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", 0)
wls = [a for a in sys.argv[1:] if not a.startswith("--")] or ["cfg3_edge"]
ev = bench.HipEvents()
for wl in wls:
    for want in ((True,) if "--only" in sys.argv else (False, True)):
        R = bench.Runner(torch, np, ops, synth, dev, wl, "hwc", "fused", want_d_src=want)
        q = bench.quick(torch, np, ev, R)
        print("%s d_src=%s: step %.2f us, main kernel %.2f us" % (wl, want, q["ms_per_step"] * 1e3, q["main_kernel_ms"] * 1e3), flush=True)
