#!/usr/bin/env python3
"""A/B timing of several builds of libsfmwarp.so in ONE process, interleaved rounds (guide rule 24).

    python tools/ab_inproc.py [--workload cfg3] [--rounds 7] [--iters 40] [--mode fused|fwd|bwd] a.so b.so ...

Every library is dlopen'ed privately (RTLD_LOCAL) and driven through the C ABI with the SAME descriptor,
inputs and outputs; each round runs `iters` back-to-back steps of every library in turn.  Prints, per
library, the median / min over the rounds of the whole-step time (HIP events around the iters) and of the main
kernel (sfm_loss_profile_events on one step per round).  Timing only: results are not checked here.
"""
import argparse
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
_lib = importlib.import_module(PKG + "._lib")
synth = importlib.import_module(PKG + ".synth")


def load(path):
    lib = C.CDLL(os.path.abspath(path), mode=os.RTLD_LOCAL)
    for name in ("sfm_loss_workspace_bytes", "sfm_loss_fwd", "sfm_loss_bwd", "sfm_loss_fwd_bwd", "sfm_loss_profile_events"):
        res, args = _lib.SYMBOLS[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    # (an experimental build may want its workspace prepared -- the round-5 in-launch finish did: profiles/r05_finish_in_launch.txt)
    init = getattr(lib, "sfm_loss_workspace_init", None)
    if init is not None:
        init.restype, init.argtypes = C.c_int, [C.POINTER(_lib.SfmLossDesc), C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--mode", default="fused", choices=["fused", "fwd", "bwd"])
    ap.add_argument("--layout", default="hwc", choices=["hwc", "planar"])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, H, W, n_src, n_scales, cfg, desc = bench.WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=n_scales, seed=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cv = (lambda a: ops.to_hwc(t(a))) if args.layout == "hwc" else t
    fl = ops.FusedLoss(**cfg).bind([cv(a) for a in d["tgt_pyr"]], [cv(a) for a in d["src_pyr"]], t(d["intrinsics"]),
                                   [t(a) for a in d["disps"]], [t(a) for a in d["poses"]], layout=args.layout)
    libs = [(os.path.basename(p), load(p)) for p in args.libs]
    loss5 = torch.zeros(5, dtype=torch.float32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    wss = {}
    for n, l in libs:      # a workspace per build, prepared by that build
        nb = l.sfm_loss_workspace_bytes(C.byref(fl.desc))
        t_ws = torch.empty((nb // 4 + 64,), dtype=torch.float32, device=dev)
        ptr = C.c_void_p(t_ws.data_ptr() + (-t_ws.data_ptr()) % 256)
        if hasattr(l, "sfm_loss_workspace_init"):
            assert l.sfm_loss_workspace_init(C.byref(fl.desc), ptr, nb, st) == 0
        wss[id(l)] = (t_ws, ptr, nb)

    def step(lib):
        _, wsp, need = wss[id(lib)]
        if args.mode == "fused":
            rc = lib.sfm_loss_fwd_bwd(C.byref(fl.desc), C.c_void_p(loss5.data_ptr()), wsp, need, st)
        elif args.mode == "fwd":
            rc = lib.sfm_loss_fwd(C.byref(fl.desc), C.c_void_p(loss5.data_ptr()), wsp, need, st)
        else:
            rc = lib.sfm_loss_bwd(C.byref(fl.desc), 1.0, wsp, need, st)
        assert rc == 0, rc

    ev = bench.HipEvents()
    e0, e1 = ev.create(), ev.create()
    res = {n: ([], [], None) for n, _ in libs}
    for n, lib in libs:   # warm-up + the loss each build reports (a sanity check of ablation builds)
        for _ in range(10):
            step(lib)
        torch.cuda.synchronize()
        res[n] = ([], [], loss5.cpu().numpy().copy())
    for r in range(args.rounds):
        for n, lib in libs:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters):
                step(lib)
            b.record()
            lib.sfm_loss_profile_events(e0, e1)
            step(lib)
            torch.cuda.synchronize()
            res[n][0].append(a.elapsed_time(b) / args.iters * 1e3)
            res[n][1].append(ev.elapsed_ms(e0, e1) * 1e3)
    base = None
    for n, _ in libs:
        stp, ker, l5 = res[n]
        med = float(np.median(ker))
        base = base or med
        print("%-34s step us: med %7.2f min %7.2f | main kernel us: med %7.2f min %7.2f (%+5.1f%% vs first) | loss5[0] %.6f" % (
            n, np.median(stp), np.min(stp), med, np.min(ker), (med / base - 1) * 100, l5[0]), flush=True)


if __name__ == "__main__":
    main()
