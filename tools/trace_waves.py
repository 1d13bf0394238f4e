#!/usr/bin/env python3
"""Diagnostics: per-wavefront timeline of the fused loss kernel (start/end, CU/SIMD placement)."""
import importlib, sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops"); synth = importlib.import_module(PKG + ".synth")
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
dev = torch.device("cuda:0")
TB, TH, TW, TS = [int(v) for v in os.environ.get("SFM_TRACE_SHAPE", "32,128,416,2").split(",")]
d = synth.make_inputs(B=TB, H=TH, W=TW, n_src=TS, n_scales=4, seed=int(os.environ.get("SFM_TRACE_SEED", "1")))
if os.environ.get("SFM_TRACE_ROLL"):   # the same samples, rotated by k: which XCD gets which sample changes, nothing else
    _k = int(os.environ["SFM_TRACE_ROLL"])
    for _key in ("tgt_pyr", "src_pyr", "disps", "poses"):
        d[_key] = [np.roll(a, _k, axis=0) for a in d[_key]]
    d["intrinsics"] = np.roll(d["intrinsics"], _k, axis=0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
if os.environ.get("SFM_TRACE_ZERO"):   # DVFS probe (MI355X_MICROARCH.md, DVFS give-back): the same launch on all-zero images
    for k in ("tgt_pyr", "src_pyr"):
        d[k] = [np.zeros_like(a) for a in d[k]]
layout = os.environ.get("SFM_LAYOUT", "hwc")
cv = (lambda a: ops.to_hwc(t(a))) if layout == "hwc" else t
fl = ops.FusedLoss(smooth_reg=0.1, ssim_rate=float(os.environ.get("SFM_TRACE_SSIM_RATE", "0.15")), smooth_mode=os.environ.get("SFM_TRACE_SMOOTH", "second_order")).bind([cv(a) for a in d["tgt_pyr"]], [cv(a) for a in d["src_pyr"]], t(d["intrinsics"]),
                                                        [t(a) for a in d["disps"]], [t(a) for a in d["poses"]], layout=layout)
run = {"fused": fl.forward_backward, "fwd": fl.forward, "bwd": lambda: fl.backward(1.0)}[mode]
for _ in range(5): run()
buf = torch.zeros((60000, 4), dtype=torch.int64, device=dev)
ops.lib.sfm_loss_debug_trace(C.c_void_p(buf.data_ptr()))
run(); torch.cuda.synchronize()
raw = buf.cpu().numpy()
nz = int((raw != 0).any(axis=1).sum())
if os.environ.get("SFM_TRACE_DUMP"):
    np.save(os.environ["SFM_TRACE_DUMP"], raw if os.environ.get("SFM_TRACE_DUMP_FULL") else raw[:nz])
stamps = os.environ.get("SFMWARP_LIB", "").endswith("stamps.so")
SW = int(os.environ.get('SFM_STAMP_WORDS', '8'))
n_items = int(os.environ['SFM_TRACE_ITEMS']) if 'SFM_TRACE_ITEMS' in os.environ else (nz // (1 + SW // 4) if stamps else nz)
a = raw[:n_items]
t0 = a[:, 0].min()
st = (a[:, 0] - t0) / 100.0; en = (a[:, 1] - t0) / 100.0   # microseconds
hw = a[:, 2]; xcc = a[:, 3] & 0xf; wg = (a[:, 3] >> 8) & 0xffffffff; wv = (a[:, 3] >> 40) & 0xff
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10 + simd
if os.environ.get("SFM_TRACE_PLACEMENT"):
    cukey = xcc * 100000 + se * 10000 + sh * 1000 + cu * 10
    for x in range(1):
        m = xcc == x
        print("XCD %d: workgroup (loc = blockIdx >> 3) -> CU placement, first 12 CUs:" % x)
        for ck in np.unique(cukey[m])[:12]:
            mm = m & (cukey == ck)
            locs = sorted(set((wg[mm] >> 3).tolist()))
            print("   CU", ck, "locs", locs, "waves per SIMD", np.bincount(simd[mm], minlength=4).tolist(),
                  " wave->simd of first wg", [(int(w_), int(s_)) for w_, s_ in zip(wv[mm & ((wg >> 3) == locs[0])], simd[mm & ((wg >> 3) == locs[0])])])
print("items", len(a), "kernel span %.1f us" % en.max(), " wave duration: mean %.1f  min %.1f  max %.1f us" % ((en - st).mean(), (en - st).min(), (en - st).max()))
print("start times: p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(st, [50, 90, 99, 100])))
u, cnt = np.unique(key, return_counts=True)
print("distinct SIMDs used", len(u), " waves per SIMD: min %d max %d mean %.2f" % (cnt.min(), cnt.max(), cnt.mean()), np.bincount(cnt))
# peak concurrency per SIMD
peak = []
for k in u[:4096]:
    m = key == k
    ev = sorted([(s, 1) for s in st[m]] + [(e, -1) for e in en[m]])
    c = 0; pk = 0
    for _, dd in ev:
        c += dd; pk = max(pk, c)
    peak.append(pk)
print("peak concurrent waves per SIMD histogram", np.bincount(peak))
late = st > 5
print("waves starting after 5us: %d ; their mean duration %.1f ; early waves mean duration %.1f" % (late.sum(), (en - st)[late].mean() if late.any() else 0, (en - st)[~late].mean()))
if os.environ.get("SFMWARP_LIB", "").endswith("stamps.so"):
    n = n_items
    b = raw.reshape(-1)[n * 4: n * 4 + n * SW].reshape(n, SW).astype(np.float64)
    steps = max(b[:, 4].sum(), 1.0)
    if os.environ.get("SFM_QUAD_STAMPS", "0") != "1": print("cycles per row step (mean over all waves): A.finish %.0f  A.issue %.0f  B %.0f  C %.0f   total %.0f ; smoothness pass per wave %.0f cycles" % (
        b[:, 0].sum() / steps, b[:, 1].sum() / steps, b[:, 2].sum() / steps, b[:, 3].sum() / steps, b[:, :4].sum() / steps, b[:, 5].mean()))
if stamps and os.environ.get("SFM_QUAD_STAMPS", "0") == "1":
    # quad kernels: b[:,0..3] = cycles waited for the partner at start-S, start-G, finish-S, finish-G; b[:,4] = steady loops
    dur_cyc = b[:, 7]
    dur_us = (a[:, 1] - a[:, 0]).astype(np.float64) / 100.0
    print("in-kernel clock: median %.3f GHz" % np.median(dur_cyc / dur_us / 1e3))
    order = np.zeros(n, int)
    for kk in np.unique(key):
        idx = np.where(key == kk)[0]
        order[idx[np.argsort(en[idx])]] = np.arange(len(idx))
    big = dur_cyc > np.percentile(dur_cyc, 30)          # the waves of the two large scales
    for rk in range(3):
        m = (order == rk) & big
        if m.any():
            print("finish-rank %d: n=%d wave %.0fk cyc = source passes %.0fk [steady loops %.0fk; waits: start-S %.1fk start-G %.1fk finish-S %.1fk finish-G %.1fk; "
                  "head %.1fk finish %.1fk pose sums %.1fk set-up %.1fk] + smoothness %.0fk + start-up/write-out %.0fk" % (
                rk, m.sum(), dur_cyc[m].mean() / 1e3, b[m, 6].mean() / 1e3, b[m, 4].mean() / 1e3, b[m, 0].mean() / 1e3, b[m, 1].mean() / 1e3,
                b[m, 2].mean() / 1e3, b[m, 3].mean() / 1e3, (b[m, 8] - b[m, 0] - b[m, 1]).mean() / 1e3, (b[m, 9] - b[m, 2] - b[m, 3]).mean() / 1e3, b[m, 10].mean() / 1e3,
                (b[m, 6] - b[m, 4] - b[m, 8] - b[m, 9] - b[m, 10]).mean() / 1e3, b[m, 5].mean() / 1e3,
                (dur_cyc[m] - b[m, 5] - b[m, 6]).mean() / 1e3))
elif stamps:
    dur_cyc = b[:, 7]                                           # whole wave, shader cycles (s_memtime)
    dur_us = (a[:, 1] - a[:, 0]).astype(np.float64) / 100.0     # the same span in 100 MHz ticks
    print("in-kernel clock (wave cycles / wave time): median %.3f GHz" % np.median(dur_cyc / dur_us / 1e3))
    loop = b[:, :4].sum(axis=1); sm = b[:, 5]; src = b[:, 6]
    order = np.zeros(n, int)
    for kk in np.unique(key):
        idx = np.where(key == kk)[0]
        order[idx[np.argsort(en[idx])]] = np.arange(len(idx))
    for rk in range(4):
        m = order == rk
        if m.any():
            print("finish-rank %d: n=%d wave %.0fk cyc = source passes %.0fk (row loop %.0fk: %.0f/step x %.1f steps; set-up + prologue + pose sums %.0fk)"
                  " + smoothness pass %.0fk + start-up/write-out %.0fk | A.fin %.0f A.iss %.0f B %.0f C %.0f per step" % (
                rk, m.sum(), dur_cyc[m].mean() / 1e3, src[m].mean() / 1e3, loop[m].mean() / 1e3, (loop[m] / b[m, 4]).mean(), b[m, 4].mean(),
                (src[m] - loop[m]).mean() / 1e3, sm[m].mean() / 1e3, (dur_cyc[m] - src[m] - sm[m]).mean() / 1e3,
                (b[m, 0] / b[m, 4]).mean(), (b[m, 1] / b[m, 4]).mean(), (b[m, 2] / b[m, 4]).mean(), (b[m, 3] / b[m, 4]).mean()))
# finish time by DISPATCH round (age rank on the SIMD: workgroups j, j + S, j + 2S of an XCD share a SIMD)
loc = wg >> 3
S_xcd = max(1, len(u) // 8)
drank = np.minimum(loc // S_xcd, 3)
for rk in range(4):
    m = drank == rk
    if m.any():
        print("dispatch-round %d: n=%d  start p50 %.2f us  end mean %.1f p10 %.1f p50 %.1f p90 %.1f us" % (
            rk, m.sum(), np.percentile(st[m], 50), en[m].mean(), *np.percentile(en[m], [10, 50, 90])))
order_d = np.zeros(len(a), int)
for kk in u:
    idx = np.where(key == kk)[0]
    order_d[idx[np.argsort(en[idx])]] = np.arange(len(idx))
print("finish order vs dispatch round (rows: dispatch round, cols: finish rank on its SIMD):")
for rk in range(3):
    print("  ", rk, [int(((drank == rk) & (order_d == f)).sum()) for f in range(3)])
# per-SIMD completion times: is the launch limited by throughput (all SIMDs end together) or by imbalance?
simd_end = np.array([en[key == kk].max() for kk in u])
simd_work = np.array([(en - st)[key == kk].sum() for kk in u])
print("SIMD completion time: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile(simd_end, [0, 10, 50, 90, 100])))
xk = (u // 100000)
for x in range(8):
    m = xk == x
    if m.any():
        print("  XCD %d: SIMDs %d, completion p50 %.1f max %.1f us, waves %d" % (x, m.sum(), np.percentile(simd_end[m], 50), simd_end[m].max(), cnt[m].sum()))
