#!/usr/bin/env python3
"""Condenses gpurun_out/profiles_raw/<tag>/ (see collect_profiles.sh) into profiles/<tag>_*.{csv,json,md}."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
raw = os.path.join(ROOT, "gpurun_out", "profiles_raw", tag)
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    g = glob.glob(os.path.join(raw, pattern))   # gpurun merges every run into the same directory: take the newest
    return max(g, key=os.path.getmtime) if g else None


stats = one("trace/*/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(out, "%s_kernel_stats.csv" % tag))
bench_line = [l for l in open(os.path.join(raw, "trace.log")) if l.startswith("{")][-1]
bench = json.loads(bench_line)


def counters(sub):
    f = one("%s/*/*_counter_collection.csv" % sub)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


pm = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):
    for k, d in counters(sub).items():
        pm.setdefault(k, {}).update(d)
main = [k for k in pm if "loss_kernel" in k]
summary = {"tag": tag, "bench": bench, "kernels": {}}
for r in csv.DictReader(open(stats)):
    summary["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
for k in main:
    c = pm[k]
    px = bench["config"]["warped_px_per_gpu_step"]
    fetch_kb, write_kb = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    t = {"counters_per_launch": c}
    if fetch_kb is not None and write_kb is not None:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 64 B per
        # 128-B request for wide coalesced streams (x2 correction, calibrated for 16 B/lane loads only); this kernel
        # loads 4-8 B per lane (uncalibrated), so both the raw and the x2-corrected figure are given.
        t["hbm_bytes_raw"] = (fetch_kb + write_kb) * 1024
        t["hbm_bytes_fetch_x2"] = (2 * fetch_kb + write_kb) * 1024
        # template arguments <SSIM, GRAD, LOSS, EXPL, SMODE>: fused = 28 + 32 B per warped px, backward 32, forward 28
        import re
        m = re.search(r"loss_kernel<(\w+), (\w+), (\w+)", k)
        grad, loss = (m.group(2) == "true", m.group(3) == "true") if m else (True, True)
        t["entry_point"] = "sfm_loss_fwd_bwd" if (grad and loss) else ("sfm_loss_bwd" if grad else "sfm_loss_fwd")
        t["algorithmic_bytes"] = ((28 if loss else 0) + (32 if grad else 0)) * px
    summary["kernels"].setdefault(k, {}).update(t)
json.dump(summary, open(os.path.join(out, "%s_summary.json" % tag), "w"), indent=1, sort_keys=True)
with open(os.path.join(out, "%s_summary.md" % tag), "w") as f:
    f.write("# rocprofv3 summary %s\n\ncommand: `python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --mode %s` under "
            "`rocprofv3 --kernel-trace --stats` (kernel durations; %d blocks of 20 timed steps) and separate `--pmc` passes (counters, 6-step runs)\n\n" % (
                tag, bench["config"]["mode"], bench.get("timing", {}).get("blocks", 1)))
    f.write("bench line inside the profiler: value %.0f %s, %.4f ms/step, dominant kernel %.2f us by HIP events\n\n" % (
        bench["value"], bench["unit"], bench["ms_per_step"], bench["roofline"]["kernel_ms"] * 1e3))
    f.write("| kernel | calls | avg us | % |\n|---|---|---|---|\n")
    for name, v in summary["kernels"].items():
        if "avg_ns" in v:
            f.write("| `%s` | %d | %.2f | %.1f |\n" % (name[:70], v["calls"], v["avg_ns"] / 1e3, v["pct"]))
    for k in main:
        v = summary["kernels"][k]
        f.write("\n## PMC, `%s` = %s (per launch)\n\n" % (k[:70], v.get("entry_point", "")))
        for c, val in sorted(v.get("counters_per_launch", {}).items()):
            f.write("* %s = %.4g\n" % (c, val))
        if "hbm_bytes_raw" in v:
            f.write("\nHBM-side traffic: raw (FETCH_SIZE + WRITE_SIZE) x 1024 = %.1f MB; with the guide's x2 FETCH correction %.1f MB; "
                    "algorithmic bytes of the launch %.1f MB\n" % (v["hbm_bytes_raw"] / 1e6, v["hbm_bytes_fetch_x2"] / 1e6, v["algorithmic_bytes"] / 1e6))
print(open(os.path.join(out, "%s_summary.md" % tag)).read())
