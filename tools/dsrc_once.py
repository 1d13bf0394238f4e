#!/usr/bin/env python3
"""A handful of fused steps with dL/d(src) bound (for rocprofv3 --pmc / --kernel-trace over the two launches): tools/dsrc_once.py [workload] [steps]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bench = importlib.import_module("bench")
PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops"); synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_edge"
R = bench.Runner(torch, np, ops, synth, dev, wl, "hwc", "fused", want_d_src=True)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    R.step()
torch.cuda.synchronize()
