#!/usr/bin/env python3
"""Time series of the cfg3 step in ONE process over ~25 s (blocks of 20 steps between synchronizes), next to the clocks the driver
exposes in sysfs, sampled by a thread: are the 57.5 / 60.5 us modes EPISODES in time?  (tools/alloc_mode_probe.py: a set of arrays
measured slow reads fast two seconds later.)

    python tools/mode_timeseries.py [seconds]
"""
import glob
import importlib
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bench = importlib.import_module("bench")
PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops")
synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", 0)
T = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0

devdir = None
for r in glob.glob("/sys/class/drm/renderD*"):
    if os.path.exists("/dev/dri/" + os.path.basename(r)):
        devdir = r + "/device"
files = {}
if devdir:
    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
        if os.path.exists(devdir + "/" + name):
            files[name] = devdir + "/" + name
    for f in glob.glob(devdir + "/hwmon/hwmon*/*_input") + glob.glob(devdir + "/hwmon/hwmon*/power1_average"):
        files[os.path.basename(f)] = f
samples, stop = [], False


def cur(name, text):
    if name.startswith("pp_dpm"):
        for line in text.splitlines():
            if line.rstrip().endswith("*"):
                return line.split(":")[1].replace("*", "").strip()
        return "?"
    return text.strip()


def sampler():
    while not stop:
        row = [time.perf_counter()]
        for name, f in files.items():
            try:
                row.append(cur(name, open(f).read()))
            except Exception:
                row.append("nan")
        samples.append(row)
        time.sleep(0.1)


R = bench.Runner(torch, np, ops, synth, dev, "cfg3_edge", "hwc", "fused")
for _ in range(200):
    R.step()
torch.cuda.synchronize()
th = threading.Thread(target=sampler, daemon=True)
th.start()
t_begin = time.perf_counter()
series = []
while time.perf_counter() - t_begin < T:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        R.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    series.append((t0 - t_begin, (t1 - t0) / 20 * 1e6))
stop = True
th.join()
a = np.array(series)
slow = a[:, 1] > 59.0
print("blocks %d  median %.2f us  p10 %.2f  p90 %.2f  share of blocks above 59 us: %.1f %%" % (len(a), np.median(a[:, 1]), np.percentile(a[:, 1], 10), np.percentile(a[:, 1], 90), 100 * slow.mean()))
# episodes
ep, start = [], None
for i, s in enumerate(slow):
    if s and start is None:
        start = i
    if not s and start is not None:
        if i - start >= 5:
            ep.append((a[start, 0], a[i, 0] - a[start, 0]))
        start = None
if start is not None:
    ep.append((a[start, 0], a[-1, 0] - a[start, 0]))
print("slow episodes (start s, duration s):", ", ".join("%.2f+%.2f" % e for e in ep) or "none")
# half-second means next to the sampled clocks
names = list(files)
print("t(s)  step_us  " + "  ".join(names))
for t in np.arange(0, T, 0.5):
    m = (a[:, 0] >= t) & (a[:, 0] < t + 0.5)
    rows = [r for r in samples if t <= r[0] - t_begin < t + 0.5]
    vals = []
    for k in range(len(names)):
        col = [r[1 + k] for r in rows]
        vals.append(max(set(col), key=col.count) if col else "-")
    if m.any():
        print("%4.1f  %6.2f  %s" % (t, a[m, 1].mean(), "  ".join(vals)))
