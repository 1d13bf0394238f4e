#!/usr/bin/env python3
"""Condenses gpurun_out/pmc_gather/* (tools/pmc_gather.sh) into profiles/r05_gather_pmc.txt: per workload, the counters of the main
kernel per launch (mean over its launches)."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, "gpurun_out", "pmc_gather")
out = []
for wl in ("cfg3_edge", "cfg3_smooth_disp", "cfg5_2src", "cfg5_2src_smooth_disp"):
    vals = collections.OrderedDict()
    for p in ("p1", "p2", "p3"):
        for f in glob.glob(os.path.join(base, "%s_%s" % (wl, p), "**", "*counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                if "loss_kernel" in row["Kernel_Name"]:
                    acc[(row["Counter_Name"], row["Dispatch_Id"])].append(float(row["Counter_Value"]))
            per = collections.defaultdict(list)
            for (name, disp), v in acc.items():
                per[name].append(sum(v))
            for name, v in per.items():
                vals[name] = sum(v) / len(v)
    out.append((wl, vals))
lines = ["rocprofv3 --pmc passes of `bench.py --steps 6 --warmup 2 --workload W` (tools/pmc_gather.sh), main kernel, per launch (mean over its launches)", ""]
for wl, v in out:
    lines.append("%-24s %s" % (wl, "  ".join("%s %.4g" % kv for kv in v.items())))
lines.append("")
for a, b in (("cfg3_edge", "cfg3_smooth_disp"), ("cfg5_2src", "cfg5_2src_smooth_disp")):
    va, vb = dict(out)[a], dict(out)[b]
    lines.append("%s / %s:  " % (a, b) + "  ".join("%s x%.2f" % (k, va[k] / vb[k]) for k in va if k in vb and vb[k]))
text = "\n".join(lines) + "\n"
open(os.path.join(ROOT, "profiles", "r05_gather_pmc.txt"), "w").write(text)
print(text)
