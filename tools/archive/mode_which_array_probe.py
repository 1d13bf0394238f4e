#!/usr/bin/env python3
"""Which array's placement sets the level of a cfg3 step?  One process: the arrays of one family (target pyramid / source pyramid /
disparities / everything the binding allocates itself) are re-allocated, everything else stays; each take is timed.  Earlier takes'
arrays stay alive, so every re-allocation gets fresh addresses.

    python tools/mode_which_array_probe.py
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


import glob
import threading
_dev = None
for _r in glob.glob("/sys/class/drm/renderD*"):
    if os.path.exists("/dev/dri/" + os.path.basename(_r)):
        _dev = _r + "/device"
_files = [(n, _dev + "/" + n) for n in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk") if _dev and os.path.exists(_dev + "/" + n)]
_files += [(os.path.basename(f), f) for f in (glob.glob(_dev + "/hwmon/hwmon*/power1_input") if _dev else [])]
clock_note = [""]


def _cur(name, text):
    if name.startswith("pp_dpm"):
        for line in text.splitlines():
            if line.rstrip().endswith("*"):
                return line.split(":")[1].replace("*", "").strip()
        return "?"
    return "%.0fW" % (float(text) / 1e6)


def take(step_fn, k=20, blocks=8):
    rows, stop = [], [False]

    def sampler():
        while not stop[0]:
            rows.append([_cur(n, open(f).read()) for n, f in _files])
            time.sleep(0.02)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    try:
        return _take(step_fn, k, blocks)
    finally:
        stop[0] = True
        th.join()
        cols = list(zip(*rows)) if rows else []
        clock_note[0] = " ".join("%s=%s" % (n.replace("pp_dpm_", ""), max(set(c), key=c.count)) for (n, _), c in zip(_files, cols))


def _take(step_fn, k=20, blocks=8):
    pair = [ev.create(), ev.create()]
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.12:
        for _ in range(50):
            step_fn(None)
        torch.cuda.synchronize()
    ts, ks = [], []
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            step_fn(pair if i == k // 2 else None)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / k * 1e6)
        ks.append(ev.elapsed_ms(pair[0], pair[1]) * 1e3)
    return float(np.median(ts)), float(np.median(ks))


B, H, W, n_src, n_scales, cfg, _ = bench.WORKLOADS["cfg3_edge"]
d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=n_scales, seed=1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
if os.environ.get("PROBE_SACRIFICE_MB"):    # a block that is allocated first, kept, and used for nothing
    sacrificed = torch.empty((int(os.environ["PROBE_SACRIFICE_MB"]) << 20,), dtype=torch.uint8, device=dev)
if os.environ.get("PROBE_FULL_FIRST"):      # what bench.Runner holds as well: the full-resolution frames, allocated BEFORE the pyramids
    full = (t(d["tgt"]), t(d["src"]))
if os.environ.get("PROBE_SRC_FIRST"):
    _src = [ops.to_hwc(t(a)) for a in d["src_pyr"]]
    _tgt = [ops.to_hwc(t(a)) for a in d["tgt_pyr"]]
else:
    _tgt = [ops.to_hwc(t(a)) for a in d["tgt_pyr"]]
    _src = [ops.to_hwc(t(a)) for a in d["src_pyr"]]
cur = dict(tgt=_tgt, src=_src, disp=[t(a) for a in d["disps"]], K=t(d["intrinsics"]), pose=[t(a) for a in d["poses"]])
keep = []


def bind():
    fl = ops.FusedLoss(**cfg).bind(cur["tgt"], cur["src"], cur["K"], cur["disp"], cur["pose"], norm_B=B, layout="hwc")
    keep.append(fl)

    def step(evs):
        if evs:
            ops.lib.sfm_loss_profile_events(evs[0], evs[1])
        fl.forward_backward()
    return step


def show(label):
    s, k = take(bind())
    print("%-40s step %.2f kernel %.2f  %s" % (label, s, k, "SLOW" if s > 58.2 else "fast"), flush=True)


show("as first allocated")
if os.environ.get("PROBE_TOUCH"):      # the SAME arrays, same binding: read (a sum), then re-written in place (x 1.0)
    step_fn = bind()
    print("%-40s step %.2f kernel %.2f" % (("same arrays, bound again",) + take(step_fn)), clock_note[0], flush=True)
    acc = 0.0
    if os.environ.get("PROBE_TOUCH") == "families":      # one family at a time, smallest first: which read ends the slow level?
        fl_last = keep[-1]
        unrelated = torch.ones((64,), dtype=torch.float32, device=dev)          # (allocated now: a new small block)
        torch.cuda.synchronize()
        print("%-40s step %.2f kernel %.2f" % (("after ALLOCATING + filling 64 floats",) + take(step_fn)), flush=True)
        if os.environ.get("PROBE_D2H"):       # the pieces of `float(x.sum())` one at a time: a reduction kernel; a device-to-host copy
            r = unrelated.sum()
            torch.cuda.synchronize()
            print("%-40s step %.2f kernel %.2f" % (("after a sum KERNEL over them (no copy)",) + take(step_fn)), flush=True)
            host = unrelated.cpu()
            print("%-40s step %.2f kernel %.2f" % (("after COPYING them to the host",) + take(step_fn)), flush=True)
            acc += float(r)
            print("%-40s step %.2f kernel %.2f" % (("after float() of the 0-d result",) + take(step_fn)), flush=True)
            sys.exit(0)
        acc += float(unrelated.sum())
        print("%-40s step %.2f kernel %.2f" % (("after summing those 64 floats",) + take(step_fn)), flush=True)
        for label, arrs in (("K + poses", [cur["K"]] + cur["pose"]), ("workspace + outputs", [fl_last.ws] + fl_last.d_disps + fl_last.d_poses),
                            ("disp", cur["disp"]), ("tgt", cur["tgt"]), ("src", cur["src"])):
            for a in arrs:
                acc += float(a.sum())
            print("%-40s step %.2f kernel %.2f" % (("after READING %s" % label,) + take(step_fn)), flush=True)
        sys.exit(0)
    for fam in ("tgt", "src", "disp"):
        for a in cur[fam]:
            acc += float(a.sum())
    print("%-40s step %.2f kernel %.2f" % (("after READING every input (sums)",) + take(step_fn)), clock_note[0], flush=True)
    for fam in ("disp", "tgt", "src"):
        for a in cur[fam]:
            a.mul_(1.0)
        torch.cuda.synchronize()
        print("%-40s step %.2f kernel %.2f" % (("after re-WRITING %s in place" % fam,) + take(step_fn)), clock_note[0], flush=True)
    sys.exit(0)
show("outputs + workspace re-allocated")
for rep in range(int(os.environ.get("PROBE_REPS", "3"))):
    for fam in os.environ.get("PROBE_ORDER", "tgt,src,disp").split(","):
        keep.append(cur[fam])
        pad = torch.empty(((rep * 3 + 1) << 20,), dtype=torch.uint8, device=dev)      # (shifts what the allocator hands out next)
        keep.append(pad)
        cur[fam] = [a.clone() for a in cur[fam]]
        show("%s re-allocated (%d)" % (fam, rep))
