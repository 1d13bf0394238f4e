#!/usr/bin/env python3
"""Condenses gpurun_out/profiles_raw/<tag>/<variant>/ (see collect_profiles.sh; variant = workload_layout_mode) into
profiles/<tag>_summary.{json,md} and one profiles/<tag>_kernel_stats_<variant>.csv per variant."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
raw_root = os.path.join(ROOT, "gpurun_out", "profiles_raw", tag)
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def summarize(raw, variant):
    def one(pattern):
        g = glob.glob(os.path.join(raw, pattern))   # gpurun merges every run into the same directory: take the newest
        return max(g, key=os.path.getmtime) if g else None

    stats = one("trace/*/*_kernel_stats.csv")
    shutil.copy(stats, os.path.join(out, "%s_kernel_stats_%s.csv" % (tag, variant)))
    bench = json.loads([l for l in open(os.path.join(raw, "trace.log")) if l.startswith("{")][-1])

    def canon(name):
        # the kernels' canonical names in every file under profiles/ and in bench.py: "...<template arguments>(sfm::LossArgs)" -- since the
        # main kernels take their header as preloaded scalar arguments in front of the struct, the profiler prints a longer parameter list
        return re.sub(r"\((?:unsigned int|int|unsigned long\*|unsigned long long\*|float const\*|, )+sfm::LossArgs(?:, float\*)?\)", "(sfm::LossArgs)", name)

    def counters(sub):
        f = one("%s/*/*_counter_collection.csv" % sub)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        if f:
            for r in csv.DictReader(open(f)):
                agg[canon(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}

    pm = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):
        for k, d in counters(sub).items():
            pm.setdefault(k, {}).update(d)
    summary = {"bench": bench, "kernels": {}}
    for r in csv.DictReader(open(stats)):
        if "sfm::" in r["Name"]:
            summary["kernels"][canon(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
    px = bench["config"]["warped_px_per_gpu_step"]
    for k in [k for k in pm if "loss_kernel" in k]:
        c = pm[k]
        t = {"counters_per_launch": c}
        fetch_kb, write_kb = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
        if fetch_kb is not None and write_kb is not None:
            # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 64 B per 128-B request
            # for wide coalesced streams (x2 correction, calibrated for 16 B/lane loads only); this kernel loads 4-24 B per lane
            # (uncalibrated), so both the raw and the x2-corrected figure are given.
            t["hbm_bytes_raw"] = (fetch_kb + write_kb) * 1024
            t["hbm_bytes_fetch_x2"] = (2 * fetch_kb + write_kb) * 1024
        m = re.search(r"loss_kernel<(\w+), (\w+), (\w+)", k)     # <SSIM, GRAD, LOSS, ...>: fused = 28 + 32 B per warped px
        grad, loss = (m.group(2) == "true", m.group(3) == "true") if m else (True, True)
        mw = re.search(r"loss_kernel_wide<(\w+)", k)             # <LOSS, SMODE, HWC>: the L1 gradient kernels' build for small launches
        if mw:
            grad, loss = True, mw.group(1) == "true"
        t["entry_point"] = "sfm_loss_fwd_bwd" if (grad and loss) else ("sfm_loss_bwd" if grad else "sfm_loss_fwd")
        t["algorithmic_bytes"] = ((28 if loss else 0) + (32 if grad else 0)) * px
        if k in summary["kernels"] and "avg_ns" in summary["kernels"][k]:
            t["roofline_frac_by_rocprof_duration"] = t["algorithmic_bytes"] / (summary["kernels"][k]["avg_ns"] * 1e-9) / 8e12
        summary["kernels"].setdefault(k, {}).update(t)
    return summary


variants = {}
for d in sorted(glob.glob(os.path.join(raw_root, "*_*_*"))):
    if os.path.exists(os.path.join(d, "trace.log")):
        try:
            variants[os.path.basename(d)] = summarize(d, os.path.basename(d))
        except Exception as e:
            print("variant %s skipped: %s: %s" % (os.path.basename(d), type(e).__name__, e))
json.dump({"tag": tag, "variants": variants}, open(os.path.join(out, "%s_summary.json" % tag), "w"), indent=1, sort_keys=True)
with open(os.path.join(out, "%s_summary.md" % tag), "w") as f:
    f.write("# rocprofv3 summaries %s\n\nPer variant (workload_layout_mode): `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary "
            "--workload W --layout L --mode M` under `rocprofv3 --kernel-trace --stats` (durations) and separate `--pmc` passes (6-step runs).\n" % tag)
    for v, s in variants.items():
        b = s["bench"]
        f.write("\n## %s\n\nbench line inside the profiler: %.0f %s, %.4f ms/step, dominant kernel %.2f us by HIP events (frac %.4f)\n\n" % (
            v, b["value"], b["unit"], b["ms_per_step"], b["roofline"]["kernel_ms"] * 1e3, b["roofline"]["frac"]))
        f.write("| kernel | calls | avg us (rocprof) | % | algorithmic bytes / rocprof duration / 8 TB/s |\n|---|---|---|---|---|\n")
        for name, k in s["kernels"].items():
            if "avg_ns" in k:
                f.write("| `%s` | %d | %.2f | %.1f | %s |\n" % (name[:80], k["calls"], k["avg_ns"] / 1e3, k["pct"],
                                                         "%.4f" % k["roofline_frac_by_rocprof_duration"] if "roofline_frac_by_rocprof_duration" in k else ""))
        for name, k in s["kernels"].items():
            if "counters_per_launch" in k:
                f.write("\nPMC per launch of `%s`: " % name[:80] + ", ".join("%s %.4g" % (c, val) for c, val in sorted(k["counters_per_launch"].items())) + "\n")
                c = k["counters_per_launch"]
                if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
                    f.write("L2 (TCC) hit rate: %.1f %% of %.3g requests\n" % (100 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), c["TCC_HIT_sum"] + c["TCC_MISS_sum"]))
                if "hbm_bytes_raw" in k:
                    f.write("HBM-side traffic: raw (FETCH_SIZE + WRITE_SIZE) x 1024 = %.1f MB; with the guide's x2 FETCH correction %.1f MB; algorithmic %.1f MB\n" % (
                        k["hbm_bytes_raw"] / 1e6, k["hbm_bytes_fetch_x2"] / 1e6, k["algorithmic_bytes"] / 1e6))
print(open(os.path.join(out, "%s_summary.md" % tag)).read())
