#!/usr/bin/env python3
"""bench.py -- throughput of the fused view-synthesis loss path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = forward + backward of the multi-scale photometric loss through the C ABI over one
synthetic batch that is already resident in HBM -- by default the single fused launch
(sfm_loss_fwd_bwd: loss and all gradients, what SFMLearnerLoss.__call__ runs when backprop is
enabled); `--mode separate` times sfm_loss_fwd followed by sfm_loss_bwd instead -- plus, for
N > 1, the RCCL all-reduce of the five reported scalars.  The workload at any N is
BASELINE.json configs[2]/[3]: B = 32 samples PER GPU, 128x416, 4 scales, 2 sources,
L1 + SSIM(0.15) + second-order smoothness(0.1) (experiments/sfm_learner_v1_ssim.yml), weak scaling.

Metric: warped output Mpixels/s, pixels := B * n_src * sum_s h_s*w_s per step.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "sfm-learner-chainer_amd"

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_FWD, BYTES_BWD = 28, 32   # algorithmic bytes per warped pixel, SURVEY.md §8(d)

WORKLOADS = {
    # name: (B per GPU, H, W, n_src, n_scales, loss config, description)
    "cfg3": (32, 128, 416, 2, 4, dict(smooth_reg=0.1, ssim_rate=0.15),
             "BASELINE cfg3: B=32/GPU, 128x416, 4 scales, 2 src, L1+SSIM(0.15)+2nd-order smoothness(0.1)"),
    "cfg2": (8, 128, 416, 2, 4, dict(smooth_reg=0.1),
             "BASELINE cfg2: B=8, 128x416, 4 scales, 2 src, L1 + smoothness"),
    "cfg3_edge": (32, 128, 416, 2, 4, dict(smooth_reg=0.1, ssim_rate=0.15, smooth_mode="edge_aware"),
                  "cfg3 with the edge-aware smoothness (base_model.py:144-155)"),
    "cfg5": (8, 256, 832, 4, 4, dict(smooth_reg=0.1, ssim_rate=0.15),
             "BASELINE cfg5: B=8, 256x832, 5-frame (4 src), 4 scales"),
    "cfg5_2src": (8, 256, 832, 2, 4, dict(smooth_reg=0.1, ssim_rate=0.15),
                  "BASELINE cfg5 as parenthesised: B=8, 256x832, 2 src, 4 scales"),
}


class HipEvents:
    """Raw hipEvent_t pairs (torch.cuda.Event does not expose a handle before its first record)."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventDestroy.argtypes = [C.c_void_p]

    def create(self):
        ev = C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(ev)) == 0
        return ev

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        assert self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return ms.value

    def destroy(self, ev):
        self.hip.hipEventDestroy(ev)


def cpu_baseline(budget_s=15.0):
    """The oracle (NumPy restatement of the reference's CPU path) timed on this box's host
    cores on BASELINE cfg1: B=1, 128x416, 1 scale, 2 sources, L1 only, forward + backward."""
    import numpy as np
    from oracle import sfm_oracle as O
    synth = importlib.import_module(PKG + ".synth")
    d = synth.make_inputs(B=1, H=128, W=416, n_src=2, n_scales=1, seed=1)

    def step():
        return O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], backward=True)

    step()
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 20 and (time.perf_counter() < t_end or len(times) < 3):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    px = 1 * 2 * 128 * 416
    out = {"value": round(px / med / 1e6, 4), "unit": "Mpix/s", "cores": 1, "kind": "port",
           "sample": "cfg1 (B=1, 128x416, 1 scale, 2 src, L1 only) fwd+bwd, median of %d runs, %.3f s/step; "
                     "single-threaded NumPy oracle, host has %d cores (%d usable)" % (
                         len(times), med, os.cpu_count(), len(os.sched_getaffinity(0)))}
    try:
        out["all_cores"] = cpu_baseline_all_cores(px)
    except Exception as e:   # the one-core figure above is the baseline; this one is a courtesy
        out["all_cores"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


_CPU_WORKER = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import importlib
import numpy as np
from oracle import sfm_oracle as O
synth = importlib.import_module(sys.argv[2] + ".synth")
d = synth.make_inputs(B=1, H=128, W=416, n_src=2, n_scales=1, seed=int(sys.argv[3]))
step = lambda: O.sfm_loss(d["tgt_pyr"], d["src_pyr"], d["intrinsics"], d["disps"], d["poses"], backward=True)
step()
sys.stdout.write("ready\n"); sys.stdout.flush()
sys.stdin.readline()                      # start gun: every worker is warm before any of them is timed
n = 0; t0 = time.perf_counter(); budget = float(sys.argv[4])
while time.perf_counter() - t0 < budget:
    step(); n += 1
sys.stdout.write("%d %.6f\n" % (n, time.perf_counter() - t0)); sys.stdout.flush()
"""


def cpu_baseline_all_cores(px_per_step, budget_s=8.0):
    """the same cfg1 step as independent samples, one single-threaded process per usable
    core (the path shards by sample), all started together; value = samples finished / wall time."""
    import subprocess
    # a 1-GPU box is given a share of 16 host cores, whatever the affinity mask says
    n = min(len(os.sched_getaffinity(0)), int(os.environ.get("SFM_CPU_BASELINE_PROCS", "16")))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    here = os.path.dirname(os.path.abspath(__file__))
    procs = [subprocess.Popen([sys.executable, "-c", _CPU_WORKER, here, PKG, str(k + 1), str(budget_s)], env=env,
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for k in range(n)]
    try:
        for p in procs:
            assert p.stdout.readline().strip() == "ready"
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        rate = 0.0
        steps = 0
        for p in procs:
            k, dt = p.stdout.readline().split()
            rate += int(k) * px_per_step / float(dt)
            steps += int(k)
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
            p.wait(timeout=60)
    return {"value": round(rate / 1e6, 4), "unit": "Mpix/s", "cores": n,
            "sample": "%d independent cfg1 samples in %d single-threaded processes, %.0f s each" % (steps, n, budget_s)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fused", choices=["separate", "fused"],
                    help="separate: sfm_loss_fwd then sfm_loss_bwd (the reference's forward / loss.backward()); "
                         "fused: one sfm_loss_fwd_bwd launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch of the workload (experiments only)")
    ap.add_argument("--layout", default="hwc", choices=["hwc", "planar"],
                    help="memory layout of the image pyramids resident in HBM when the timed region starts: hwc = pixel-"
                         "interleaved, as sfm_pyramid_hwc_fwd writes them (default); planar = the reference's (B,3,h,w)")
    ap.add_argument("--report-interval", type=int, default=0,
                    help="steps per reporting interval: the five scalars of the steps of an interval are summed over the ranks "
                         "with ONE all-reduce at its end (the reference's trainer reports at LogReport's interval, not per "
                         "iteration).  0 = one interval over the K timed steps; 1 = a collective every step")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ     # under torch.distributed.run, also at N = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ops = importlib.import_module(PKG + ".ops")
    synth = importlib.import_module(PKG + ".synth")
    B, H, W, n_src, n_scales, cfg, desc = WORKLOADS[args.workload]
    if args.batch > 0:
        B, desc = args.batch, desc + " [per-GPU batch overridden to %d]" % args.batch
    d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=n_scales, seed=1 + rank)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tgt_planar, src_planar = [t(a) for a in d["tgt_pyr"]], [t(a) for a in d["src_pyr"]]
    common = (t(d["intrinsics"]), [t(a) for a in d["disps"]], [t(a) for a in d["poses"]])

    def bound(layout):
        if layout == "hwc":     # the same values, pixel-interleaved
            return ops.FusedLoss(**cfg).bind([ops.to_hwc(a) for a in tgt_planar], [ops.to_hwc(a) for a in src_planar], *common,
                                             norm_B=B * world, layout="hwc")
        return ops.FusedLoss(**cfg).bind(tgt_planar, src_planar, *common, norm_B=B * world)

    fl = bound(args.layout)
    warped_px = B * n_src * sum((H >> s) * (W >> s) for s in range(n_scales))

    ev = HipEvents()
    lib = ops.lib

    # The only collective of the path: the five reported scalars summed over the shards (RCCL over xGMI).  Every
    # step writes its scalars into its own row of a device-resident log; the rows of a reporting interval are
    # reduced with one all-reduce at the interval's end, in stream order, inside the timed region.  (Per-sample
    # gradients never leave their rank.  A collective per step is --report-interval 1.)
    interval = args.report_interval if args.report_interval > 0 else max(args.steps, 1)
    n_log = max(args.steps, args.warmup, 1)
    loss_log = torch.zeros((n_log, 5), dtype=torch.float32, device=dev)
    rows = [loss_log[k] for k in range(n_log)]

    def step(k, evs=None):
        if args.mode == "fused":
            if evs:
                lib.sfm_loss_profile_events(evs[0], evs[1])
            fl.forward_backward(out=rows[k])
        else:
            if evs:
                lib.sfm_loss_profile_events(evs[0], evs[1])
            fl.forward(out=rows[k])
            if evs:
                lib.sfm_loss_profile_events(evs[2], evs[3])
            fl.backward(1.0)
        if use_dist and ((k + 1) % interval == 0 or k + 1 == n_steps_now[0]):
            lo = (k // interval) * interval
            dist.all_reduce(loss_log[lo:k + 1])

    n_steps_now = [args.warmup]
    for k in range(args.warmup):
        step(k)
    n_steps_now[0] = args.steps
    events = [[ev.create() for _ in range(4)] for _ in range(args.steps)]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, events[k])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # secondary measurements (NOT part of the timed K steps above), for the record: the other image layout ...
    other_layout = "planar" if args.layout == "hwc" else "hwc"
    other_layout_ms = None
    if world == 1:
        fl2 = bound(other_layout)
        run2 = fl2.forward_backward if args.mode == "fused" else (lambda: (fl2.forward(), fl2.backward(1.0)))
        for _ in range(5):
            run2()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_other = max(10, args.steps // 4)
        for _ in range(n_other):
            run2()
        torch.cuda.synchronize()
        other_layout_ms = (time.perf_counter() - t1) / n_other * 1e3
        del fl2
    # ... and the other launch mode
    other_mode = "separate" if args.mode == "fused" else "fused"
    other_ms = None
    if world == 1:
        def other_step():
            if other_mode == "fused":
                fl.forward_backward()
            else:
                fl.forward()
                fl.backward(1.0)
        for _ in range(5):
            other_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_other = max(10, args.steps // 4)
        for _ in range(n_other):
            other_step()
        torch.cuda.synchronize()
        other_ms = (time.perf_counter() - t1) / n_other * 1e3

    # per-launch duration of the main kernels, from the HIP events recorded inside the timed region
    k_fwd = float(np.mean([ev.elapsed_ms(e[0], e[1]) for e in events]))
    k_bwd = float(np.mean([ev.elapsed_ms(e[2], e[3]) for e in events])) if args.mode == "separate" else None
    loss = loss_log[max(args.steps - 1, 0)].cpu().numpy().tolist()
    for e4 in events:
        for e in e4:
            ev.destroy(e)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = warped_px * world / (elapsed / args.steps) / 1e6
        if args.mode == "fused":
            kname, kbytes, kms = "loss_kernel<grad+loss> (sfm_loss_fwd_bwd)", BYTES_FWD + BYTES_BWD, k_fwd
        else:
            kname, kbytes, kms = "loss_kernel<grad> (sfm_loss_bwd)", BYTES_BWD, k_bwd
        achieved = kbytes * warped_px / (kms * 1e-3) / 1e9
        # HBM-side bytes per launch of that kernel from the rocprofv3 PMC passes (collected offline with
        # tools/collect_profiles.sh; bench.py itself cannot read hardware counters)
        traffic = traffic_detail = None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_summary.json")))
            if prof["bench"]["config"]["mode"] == args.mode and args.workload == "cfg3" and args.batch == 0 \
                    and prof["bench"]["config"].get("image_layout", "planar") == args.layout:
                want = "sfm_loss_fwd_bwd" if args.mode == "fused" else "sfm_loss_bwd"
                for kn, kv in prof["kernels"].items():
                    is_hwc = kn.rstrip().endswith("true>(sfm::LossArgs)")      # last template argument: HWC
                    if kv.get("entry_point") == want and "hbm_bytes_raw" in kv and is_hwc == (args.layout == "hwc"):
                        # bytes per launch with the guide's gfx950 correction (FETCH_SIZE x 2); raw value alongside
                        traffic = round(kv["hbm_bytes_fetch_x2"])
                        traffic_detail = {"bytes_per_launch_raw": round(kv["hbm_bytes_raw"]),
                                          "bytes_per_launch_fetch_x2": round(kv["hbm_bytes_fetch_x2"]),
                                          "source": "profiles/r01_summary.json (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; "
                                                    "KiB units; FETCH_SIZE doubled per MI355X_MICROARCH.md, uncalibrated for 4-8 B/lane loads)"}
        except Exception:
            traffic = traffic_detail = None
        out = {
            "metric": "warp+photo-loss fwd+bwd Mpixels/s @128x416x4scales; % HBM roofline",
            "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "per_gpu_batch": B, "global_batch": B * world, "H": H, "W": W, "n_src": n_src,
                       "n_scales": n_scales, "mode": args.mode, "image_layout": args.layout, "warped_px_per_gpu_step": warped_px,
                       "parallelism": "batch-sharded x%d, RCCL all-reduce of the 5 scalars per reporting interval (%d steps)" % (world, interval)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_detail": traffic_detail, "kernel": kname, "kernel_ms": round(kms, 5), "bytes_per_warped_px": kbytes},
            "kernel_ms": {"fwd_main": round(k_fwd, 5) if args.mode == "separate" else None,
                          "bwd_main": round(k_bwd, 5) if k_bwd is not None else None,
                          "fused_main": round(k_fwd, 5) if args.mode == "fused" else None},
            "step_roofline_frac": round((BYTES_FWD + BYTES_BWD) * warped_px / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "other_mode": {"mode": other_mode, "ms_per_step": round(other_ms, 5) if other_ms else None,
                           "value": round(warped_px / (other_ms * 1e-3) / 1e6, 1) if other_ms else None},
            "other_layout": {"image_layout": other_layout, "ms_per_step": round(other_layout_ms, 5) if other_layout_ms else None,
                             "value": round(warped_px / (other_layout_ms * 1e-3) / 1e6, 1) if other_layout_ms else None},
            "loss5": [round(v, 6) for v in loss],
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
