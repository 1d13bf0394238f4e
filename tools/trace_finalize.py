#!/usr/bin/env python3
"""Diagnostics: where the END of a step goes -- the in-launch reduction of the main kernel (wave_arrive / finish_sample,
csrc/sfm_loss.hip).  Needs a -DSFM_FIN_STAMPS build (make -C sfm-learner-chainer_amd/csrc finstamps; SFMWARP_LIB=...).

    SFMWARP_LIB=sfm-learner-chainer_amd/libsfmwarp_finstamps.so python tools/trace_finalize.py [workload=cfg3_edge]

Stamps are 100 MHz ticks.  Per sample: the end of its last wave's work (before that wave's stores are drained and it signs in) and
the stamps of the wave that finishes the sample, all relative to the end of the LAST wave of the launch.
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "sfm-learner-chainer_amd"
bench = importlib.import_module("bench")
ops = importlib.import_module(PKG + ".ops")
synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_edge"
B, H, W, n_src, n_scales, cfg, desc = bench.WORKLOADS[wl]
d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=n_scales, seed=1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
cv = lambda a: ops.to_hwc(t(a))
fl = ops.FusedLoss(**cfg).bind([cv(a) for a in d["tgt_pyr"]], [cv(a) for a in d["src_pyr"]], t(d["intrinsics"]),
                               [t(a) for a in d["disps"]], [t(a) for a in d["poses"]], layout="hwc")
info = (C.c_int * (1 + 4 * n_scales))()
assert ops.lib.sfm_loss_plan_info(C.byref(fl.desc), 1, 1, info, len(info)) == 0
items = info[0]
tiles = [info[1 + 4 * s + 3] for s in range(n_scales)]
sample_of = np.concatenate([np.repeat(np.arange(B), tl) for tl in tiles])     # item id -> sample (items: scale-major, then sample)
assert len(sample_of) == items
for _ in range(5):
    fl.forward_backward()
buf = torch.zeros((60000, 4), dtype=torch.int64, device=dev)
NAMES = ["signed in last (add returned)", "loads issued", "rotation + K table", "loss sums stored, launch counter add issued",
         "pose sums folded", "d_pose stored", "launch counter add returned", "loss5 stored"]
for rep in range(3):
    buf.zero_()
    ops.lib.sfm_loss_debug_trace(C.c_void_p(buf.data_ptr()))      # (the hook holds for the next launch only)
    fl.forward_backward()
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(-1)
    it = raw[:items * 4].reshape(-1, 4)
    t_end_all = int(it[:, 1].max())
    t_beg = int(it[:, 0].min())
    st = raw[200000:200000 + 16 * B].reshape(B, 16)
    us = lambda v: (int(v) - t_end_all) / 100.0 if v else float("nan")
    print("== %s, launch %d: first wave start %.2f us before the last wave's end of work" % (wl, rep, (t_end_all - t_beg) / 100.0))
    last_of = np.array([it[sample_of == b, 1].max() for b in range(B)])
    rows = []
    for b in range(B):
        rows.append([us(last_of[b])] + [us(v) for v in st[b, :8]])
    rows = np.array(rows)
    order = np.argsort(rows[:, 0])
    show = list(order[:2]) + list(order[-4:]) if B > 6 else list(order)
    print("   sample: last wave's work ends | " + " | ".join(NAMES))
    for b in show:
        print("   %3d: %+7.2f | " % (b, rows[b, 0]) + " | ".join("%+7.2f" % v for v in rows[b, 1:]))
    d_chain = rows[:, 1] - rows[:, 0]
    print("   work end -> signed in (store drain + returning add): median %.2f us, max %.2f" % (np.nanmedian(d_chain), np.nanmax(d_chain)))
    print("   signed in -> d_pose stored: median %.2f us;  -> loss sums stored: median %.2f us" % (
        np.nanmedian(rows[:, 6] - rows[:, 1]), np.nanmedian(rows[:, 4] - rows[:, 1])))
    fin = np.nanmax(rows[:, 8]) if np.isfinite(rows[:, 8]).any() else float("nan")
    print("   loss5 stored %+.2f us after the last wave's end of work; last d_pose %+.2f" % (fin, np.nanmax(rows[:, 6])))
