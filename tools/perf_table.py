#!/usr/bin/env python3
"""The performance table of DESIGN.md 5, generated from the files a collection leaves under profiles/ (never typed by hand):

    python tools/perf_table.py [tag]     ->  profiles/<tag>_perf_table.md  (and the same on stdout)

Inputs: profiles/<tag>_bench_default.json (one `python bench.py` line: headline + secondary keys), profiles/<tag>_bench_driver_args.json
(`--steps 20 --warmup 5`, the driver's arguments), profiles/<tag>_summary.json (rocprofv3 durations of the same commands).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


b = last_json(P("bench_default.json"))
drv = last_json(P("bench_driver_args.json")) if os.path.exists(P("bench_driver_args.json")) else None
summ = json.load(open(P("summary.json"))) if os.path.exists(P("summary.json")) else {"variants": {}}


def rocprof_us(variant, grad=True, loss=True):
    v = summ["variants"].get(variant)
    if not v:
        return None
    want = "sfm_loss_fwd_bwd" if (grad and loss) else ("sfm_loss_bwd" if grad else "sfm_loss_fwd")
    best = None
    for name, k in v["kernels"].items():
        if "loss_kernel" in name and k.get("entry_point", want) == want and "avg_ns" in k:
            best = max(best or 0.0, k["avg_ns"] / 1e3)
    return best


def fin_us(variant):
    v = summ["variants"].get(variant)
    if not v:
        return None
    for name, k in v["kernels"].items():
        if "finalize_kernel" in name and "avg_ns" in k:
            return k["avg_ns"] / 1e3
    return None


rows = []
px = b["config"]["warped_px_per_gpu_step"]


def row(label, ms, value, k_ev_us, k_rp_us, frac_ev, step_frac, bytes_px=60, npx=px):
    frac_rp = bytes_px * npx / (k_rp_us * 1e-6) / 8e12 if k_rp_us else None
    rows.append("| %s | %.4f | %s | %s / %s | %s / %s | %s |" % (
        label, ms, ("%.0f" % value) if value else "-", ("%.1f" % k_ev_us) if k_ev_us else "-", ("%.1f" % k_rp_us) if k_rp_us else "-",
        ("%.3f" % frac_ev) if frac_ev else "-", ("%.3f" % frac_rp) if frac_rp else "-", ("%.3f" % step_frac) if step_frac else "-"))


wl = b["config"]["workload"].split(":")[0]
row("**%s**, fused, %s -- headline" % (wl, b["config"]["image_layout"]), b["ms_per_step"], b["value"], b["roofline"]["kernel_ms"] * 1e3,
    rocprof_us("cfg3_edge_hwc_fused"), b["roofline"]["frac"], b["step_roofline_frac"])
if drv:
    row("... with the driver's `--steps 20 --warmup 5`", drv["ms_per_step"], drv["value"], drv["roofline"]["kernel_ms"] * 1e3, None, drv["roofline"]["frac"], drv["step_roofline_frac"])
NAMES = [("cfg3", "cfg3, 2nd-order smoothness (the reference's live code)", "cfg3_hwc_fused"),
         ("cfg3_large_motion", "cfg3 as written, large-motion inputs", None),
         ("cfg3_smooth_disp", "cfg3 as written on a SMOOTH disparity field (logit low-passed at 1/32 of the resolution, no per-pixel noise)", None),
         ("other_layout", "cfg3 as written, planar layout", "cfg3_edge_planar_fused"),
         ("other_mode", "cfg3 as written, separate fwd + bwd", "cfg3_edge_hwc_separate"),
         ("cfg2", "cfg2 B=8, L1 + smoothness", "cfg2_hwc_fused"),
         ("cfg5", "cfg5 B=8 256x832, 4 src", "cfg5_hwc_fused"),
         ("cfg5_2src", "cfg5 as parenthesised (2 src)", "cfg5_2src_hwc_fused"),
         ("cfg5_2src_smooth_disp", "cfg5 as parenthesised (2 src) on the smooth disparity field", None),
         ("cfg1", "cfg1 B=1, 1 scale, L1", None),
         ("ref_b4", "the reference's training regime: B=4, 4 scales, L1 only (`sfm_learner_v1.yml`)", "ref_b4_hwc_fused"),
         ("cfg3_d_src", "cfg3 as written with the optional dL/d(src) bound (two launches; the kernel columns: the first one)", None),
         ("cfg3_d_src_smooth_disp", "... the same on the smooth disparity field", None)]
for key, label, variant in NAMES:
    q = b.get(key)
    if not isinstance(q, dict) or "ms_per_step" not in q:
        continue
    sep = q.get("mode") == "separate"
    npx = q["value"] * q["ms_per_step"] * 1e3 if q.get("value") else px
    row(label, q["ms_per_step"], q["value"], q["main_kernel_ms"] * 1e3, rocprof_us(variant, True, not sep) if variant else None,
        q["roofline_frac"], q["step_roofline_frac"], 32 if sep else 60, round(npx))
out = ["| workload | ms/step | Mpix/s | dominant kernel us: HIP events / rocprof avg | kernel fraction of 8 TB/s: by events / by rocprof | whole step fraction |",
       "|---|---|---|---|---|---|"] + rows
extra = []
for key, label in (("graph_ms_per_step", "the headline step replayed from a HIP graph"), ("link_ms_per_step", "through the drop-in link `SFMLearnerLoss` (pyramids + loss + backward), cfg3 as written"),
                   ("ref_b4_link_ms_per_step", "through the link at the reference's training batch (B=4, L1 only)"),
                   ("ref_b4_link_graph_ms_per_step", "... the same with `SFMLearnerLoss(use_graph=True)` (HIP-graph replay of the call)")):
    if isinstance(b.get(key), (int, float)):
        extra.append("| %s | %.4f | - | - | - | - |" % (label, b[key]))
out += extra
# round 6: the level BEFORE bench.py's one-off read of the inputs (the protocol of rounds 1-4), and the step with an unrelated kernel
# in front of every launch (the shape of a training loop)
frac = lambda ms: 60.0 * px / (ms * 1e-3) / 8e12
for line, lab in ((b, "default run"), (drv, "the driver's arguments")):
    if line and line.get("pre_warm_ms_per_step"):
        out.append("| headline BEFORE the input warm read (%s) | %.4f | %.0f | - | - | %.3f |" % (
            lab, line["pre_warm_ms_per_step"], px / (line["pre_warm_ms_per_step"] * 1e-3) / 1e6, frac(line["pre_warm_ms_per_step"])))
il = b.get("cfg3_interleaved")
if isinstance(il, dict) and "ms_per_step" in il:
    net = il["ms_per_step"] - il["extra_kernel_ms"]
    out.append("| headline with a 64-float reduction kernel in front of EVERY step (`cfg3_interleaved`; the reduction alone: %.4f ms) | %.4f | %.0f | - | - | %.3f (%.3f net of the reduction) |" % (
        il["extra_kernel_ms"], il["ms_per_step"], il["value"], il["step_roofline_frac"], frac(net)))
f = fin_us("cfg3_edge_hwc_fused")
tail = "\n\n(`finalize_kernel` by rocprof: %s us at the headline.  bench line: `csrc_sha16` %s; roofline.traffic %s; cpu_baseline %s %s on %s core(s).)\n" % (
    ("%.2f" % f) if f else "n/a", b.get("csrc_sha16"), b["roofline"].get("traffic"), b.get("cpu_baseline", {}).get("value"), b.get("cpu_baseline", {}).get("unit"),
    b.get("cpu_baseline", {}).get("cores"))
text = "\n".join(out) + tail
open(P("perf_table.md"), "w").write("# Performance table (generated by tools/perf_table.py from profiles/%s_bench_*.json and %s_summary.json)\n\n" % (tag, tag) + text)
print(text)
