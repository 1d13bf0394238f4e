#!/usr/bin/env python3
"""bench.py -- throughput of the fused view-synthesis loss path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = forward + backward of the multi-scale photometric loss through the C ABI over one synthetic batch
that is already resident in HBM: the fused launch sfm_loss_fwd_bwd (loss and all gradients -- what
SFMLearnerLoss.__call__ runs when backprop is enabled).  The path shards over samples with no exchange on the data
path; under a launcher (any N, N = 1 included) the five REPORTED scalars of EVERY step are summed over the ranks by one
ncclAllReduce(sum, fp32, count=5) issued straight through librccl on the stream the loss kernels run on (SURVEY.md 8(e);
sfm-learner-chainer_amd/rccl.py), inside the timed region.  Two other placements are measured in the same N > 1 run and reported
as secondary keys: `interval_variant` (ONE all-reduce of the K rows per block of K steps) and `per_step_torch_variant` (the
per-step all-reduce through torch.distributed: 42 us of host time per call).  Workload at any N: BASELINE.json configs[2]/[3] AS WRITTEN -- B = 32 samples PER GPU,
128x416, 4 scales, 2 sources, L1 + SSIM(0.15, experiments/sfm_learner_v1_ssim.yml) + EDGE-AWARE smoothness(0.1)
(models/base_model.py:144-155), weak scaling.  The same with the second-order smoothness the reference's live code runs
(base_model.py:75-77,169-185) is the secondary key `cfg3` of the line.

`--gpus N` with N > 1 and no torchrun environment: this process starts N ranks itself (as child processes, before
anything touches a GPU) and returns their exit code.

Timing: W untimed warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both
sides (max over ranks); blocks are repeated until >= --min-time seconds have been timed, `ms_per_step` is the MEDIAN
block (p10 / p90 / first block alongside), `value` follows from it.
Metric: warped output Mpixels/s, pixels := B * n_src * sum_s h_s*w_s per step.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "sfm-learner-chainer_amd"

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_FWD, BYTES_BWD = 28, 32   # algorithmic bytes per warped pixel, SURVEY.md 8(d)
N_SIMD = 1024                   # 256 CUs x 4 SIMDs
PROFILE_TAGS = ("r06", "r05", "r04", "r03")         # profiles/<tag>_summary.json, _issue_model.json, _wave_stage_stamps.txt: what roofline_valu is built from
GRAD_BUFFER_FLOATS = 36489060 + 3393892   # DispNet + PoseNet parameters: the ~160 MB all-reduce probe

COLLECTIVE_NOTES = {
    "step": "after EVERY step: ncclAllReduce(sum, fp32, count=5) straight through librccl on the stream of the loss kernels (SURVEY.md 8(e))",
    "interval": "ONE all-reduce of the K rows per block of K steps, in the timed region (reporting per LogReport interval)",
    "step_torch": "after EVERY step through torch.distributed.all_reduce (blocking in stream order; 42 us of host time per call)",
}
VARIANT_KEYS = {"step": "per_step_collective_variant", "interval": "interval_variant", "step_torch": "per_step_torch_variant"}

INPUT_WARM = True       # set by main from --no-input-warm: quick() follows the headline's choice
SMOOTH_DISP = dict(disp_div=32, disp_noise=0.0)
SYNTH_KW = {"cfg3_large_motion": dict(rot_sigma=0.05, trans_sigma=0.10),   # everything else: synth.make_inputs' defaults
            "cfg3_smooth_disp": SMOOTH_DISP, "cfg5_2src_smooth_disp": SMOOTH_DISP}

WORKLOADS = {
    # name: (B per GPU, H, W, n_src, n_scales, loss config, description)
    "cfg3": (32, 128, 416, 2, 4, dict(smooth_reg=0.1, ssim_rate=0.15),
             "BASELINE cfg3 with the reference's LIVE smoothness: B=32/GPU, 128x416, 4 scales, 2 src, L1+SSIM(0.15)+2nd-order smoothness(0.1)"),
    "cfg1": (1, 128, 416, 2, 1, dict(),
             "BASELINE cfg1: B=1, 128x416, 1 scale, 2 src, L1 only (the CPU baseline's workload)"),
    "cfg2": (8, 128, 416, 2, 4, dict(smooth_reg=0.1),
             "BASELINE cfg2: B=8, 128x416, 4 scales, 2 src, L1 + smoothness"),
    "l1_b32": (32, 128, 416, 2, 4, dict(smooth_reg=0.1),
               "cfg2's loss (L1 + smoothness) at B=32: the L1 kernels at full occupancy (development: tools/ab_inproc.py)"),
    "cfg3_edge": (32, 128, 416, 2, 4, dict(smooth_reg=0.1, ssim_rate=0.15, smooth_mode="edge_aware"),
                  "BASELINE cfg3 as written: B=32/GPU, 128x416, 4 scales, 2 src, L1+SSIM(0.15)+EDGE-AWARE smoothness(0.1) (base_model.py:144-155)"),
    "cfg3_large_motion": (32, 128, 416, 2, 4, dict(smooth_reg=0.1, ssim_rate=0.15, smooth_mode="edge_aware"),
                          "BASELINE cfg3 as written on LARGE-MOTION inputs: poses N(0, 0.05^2) rad / N(0, 0.1^2) instead of synth's (0.01, 0.02) -- gather "
                          "footprints several times wider, more than half of the warped pixels out of view (tests: MOTION medium)"),
    "cfg5": (8, 256, 832, 4, 4, dict(smooth_reg=0.1, ssim_rate=0.15),
             "BASELINE cfg5: B=8, 256x832, 5-frame (4 src), 4 scales"),
    "cfg5_2src": (8, 256, 832, 2, 4, dict(smooth_reg=0.1, ssim_rate=0.15),
                  "BASELINE cfg5 as parenthesised: B=8, 256x832, 2 src, 4 scales"),
    "cfg3_smooth_disp": (32, 128, 416, 2, 4, dict(smooth_reg=0.1, ssim_rate=0.15, smooth_mode="edge_aware"),
                         "BASELINE cfg3 as written on a SMOOTH disparity field (logit low-passed at 1/32 of the resolution, no per-pixel noise, "
                         "instead of synth's default 1/4 + 0.1 N(0,1): neighbouring samples' taps stay in neighbouring texels)"),
    "cfg5_2src_smooth_disp": (8, 256, 832, 2, 4, dict(smooth_reg=0.1, ssim_rate=0.15),
                              "BASELINE cfg5 as parenthesised (B=8, 256x832, 2 src) on the SMOOTH disparity field"),
    "ref_b4": (4, 128, 416, 2, 4, dict(),
               "the regime the reference TRAINS in: B=4 (experiments/sfm_learner_v1.yml:43 train_batchsize), 128x416, 4 scales, 2 src, its live "
               "loss (smooth_reg 0, exp_reg 0: L1 only, :14-16)"),
}


def csrc_sha16():
    """Content hash of the kernel sources this run was built from (the GPU box has no .git): stored with every profile collection,
    so that figures read from profiles/ can be told from figures of another build."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, PKG, "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode() + b"\0" + open(os.path.join(d, name), "rb").read() + b"\0")
    h.update(open(os.path.join(ROOT, "include", "sfmwarp.h"), "rb").read())
    return h.hexdigest()[:16]


class HipEvents:
    """Raw hipEvent_t pairs (torch.cuda.Event does not expose a handle before its first record)."""

    def __init__(self):
        # the HIP runtime instance torch (and libsfmwarp.so) already use: events of another copy of the
        # library would belong to a different runtime
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64.so" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError("libamdhip64.so is not loaded yet: import torch before creating HipEvents")
        self.hip = C.CDLL(path)
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventDestroy.argtypes = [C.c_void_p]

    def create(self):
        ev = C.c_void_p()
        rc = self.hip.hipEventCreate(C.byref(ev))
        if rc != 0:
            raise RuntimeError("hipEventCreate failed: %d" % rc)
        return ev

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        rc = self.hip.hipEventElapsedTime(C.byref(ms), a, b)
        if rc != 0:
            raise RuntimeError("hipEventElapsedTime failed: %d (was the kernel between the events launched?)" % rc)
        return ms.value

    def destroy(self, ev):
        self.hip.hipEventDestroy(ev)


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle on the host cores (rank 0, N = 1 only)
# ------------------------------------------------------------------------------------------------
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
    # a 1-GPU box is given a share of 16 host cores, whatever the affinity mask says
    n = min(len(os.sched_getaffinity(0)), int(os.environ.get("SFM_CPU_BASELINE_PROCS", "16")))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-c", _CPU_WORKER, ROOT, PKG, str(k + 1), str(budget_s)], env=env,
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


# ------------------------------------------------------------------------------------------------
# multi-GPU: start the ranks (the parent never touches a GPU)
# ------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC; without it RCCL's cross-process buffer
    # registration fails with `hipIpcGetMemHandle: invalid argument` (environment note of the task; already exported on the
    # boxes -- kept here so that a shell that lost it still works).  The children inherit everything else unchanged.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def collective_line(coll, has_comm, rehearse, comm_note):
    """config.collective of the JSON line: what issued the per-step collective -- and, loudly, when that was the fallback."""
    direct = has_comm or rehearse or coll == "step_torch"
    return COLLECTIVE_NOTES[coll] + ("" if direct else
                                     " -- FALLBACK: issued through torch.distributed.all_reduce, the direct communicator could not be made: %s" % comm_note)


def dry_run_rank(args):
    """SFM_BENCH_DRYRUN=1 (tests/test_bench_spawn_cpu.py): everything of the multi-rank plumbing EXCEPT the GPU work -- the
    rendezvous the ranks were started with (gloo instead of RCCL), the per-step collective on a 5-float row, the barrier + MAX
    reduction of the elapsed time, and the single JSON line of rank 0 -- so that `python bench.py --gpus N` can be exercised on a
    machine without GPUs.  SFM_BENCH_DRYRUN_FAIL_RANK=k makes rank k exit with code 3 (exit-code propagation)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d, but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("SFM_BENCH_DRYRUN_FAIL_RANK") == str(rank):
        sys.exit(3)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    row = torch.full((5,), float(rank + 1))
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(max(args.steps, 1)):
        step_row = row.clone()
        dist.all_reduce(step_row)
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "sum_of_ranks": float(step_row[0]),
                          "max_elapsed_s": float(tt.item())}), flush=True)
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# one measured configuration
# ------------------------------------------------------------------------------------------------
class Runner:
    """Inputs of one workload resident on the device and a bound FusedLoss; `block(K)` = K back-to-back steps."""

    def __init__(self, torch, np, ops, synth, dev, workload, layout="hwc", mode="fused", batch=0, seed=1, norm_scale=1, want_d_src=False):
        B, H, W, n_src, n_scales, cfg, desc = WORKLOADS[workload]
        if batch > 0:
            B, desc = batch, desc + " [per-GPU batch overridden to %d]" % batch
        self.torch, self.ops, self.dev = torch, ops, dev
        self.B, self.H, self.W, self.n_src, self.n_scales, self.cfg, self.desc = B, H, W, n_src, n_scales, cfg, desc
        self.layout, self.mode = layout, mode
        d = synth.make_inputs(B=B, H=H, W=W, n_src=n_src, n_scales=n_scales, seed=seed, **SYNTH_KW.get(workload, {}))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.full = (t(d["tgt"]), t(d["src"]))                         # full-resolution frames (the link's inputs)
        tgt, src = [t(a) for a in d["tgt_pyr"]], [t(a) for a in d["src_pyr"]]
        if layout == "hwc":     # the same values, pixel-interleaved
            tgt, src = [ops.to_hwc(a) for a in tgt], [ops.to_hwc(a) for a in src]
        if os.environ.get("SFM_BENCH_CLONE_INPUTS"):      # experiment (profiles/r05_process_modes.txt 1.): the pyramids copied once into fresh arrays
            fam = os.environ["SFM_BENCH_CLONE_INPUTS"]
            if "t" in fam:
                tgt = [a.clone() for a in tgt]
            if "s" in fam:
                src = [a.clone() for a in src]
        self.common = (t(d["intrinsics"]), [t(a) for a in d["disps"]], [t(a) for a in d["poses"]])
        self.fl = ops.FusedLoss(**cfg).bind(tgt, src, *self.common, norm_B=B * norm_scale, layout=layout, want_d_src=want_d_src)
        self.warped_px = B * n_src * sum((H >> s) * (W >> s) for s in range(n_scales))

    def step(self, out=None, evs=None):
        lib, fl = self.ops.lib, self.fl
        if self.mode == "fused":
            if evs:
                lib.sfm_loss_profile_events(evs[0], evs[1])
            fl.forward_backward(out=out)
        else:
            if evs:
                lib.sfm_loss_profile_events(evs[0], evs[1])
            fl.forward(out=out)
            if evs:
                lib.sfm_loss_profile_events(evs[2], evs[3])
            fl.backward(1.0)


def warm_inputs(runner):
    """Reads every input array of a bound step once with another kernel (see main: the memory-side warm-up of a measurement)."""
    keep = runner.fl._keep
    acc = 0.0
    for fam in (keep[0], keep[1], keep[3]):          # target pyramid, source pyramid, disparities
        for a in fam:
            acc += float(a.sum())
    return acc


def quick(torch, np, ev, runner, min_time=0.12, k=25):
    """Secondary measurement (NOT the headline): blocks of k steps until min_time, median block; kernel times from events."""
    pairs = [[ev.create() for _ in range(4)] for _ in range(k)]
    for _ in range(4):
        runner.step()
    if INPUT_WARM:
        warm_inputs(runner)
    for _ in range(8):
        runner.step()
    torch.cuda.synchronize()
    blocks, kt, kt2 = [], [], []
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < min_time or len(blocks) < 3:
        t0 = time.perf_counter()
        for i in range(k):
            # kernel events on ONE step per block, the one in its middle (see main: an event-carrying dispatch costs its step ~5 us,
            # which on every fifth step of a 19 us cfg2 step was +1 us on the block's mean)
            runner.step(evs=pairs[i] if i == k // 2 else None)
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / k)
        kt.append(ev.elapsed_ms(pairs[k // 2][0], pairs[k // 2][1]))
        if runner.mode != "fused":
            kt2.append(ev.elapsed_ms(pairs[k // 2][2], pairs[k // 2][3]))
    for p in pairs:
        for e in p:
            ev.destroy(e)
    ms = float(np.median(blocks)) * 1e3
    kms = float(np.mean(kt2 if kt2 else kt))
    kbytes = (BYTES_FWD + BYTES_BWD) if runner.mode == "fused" else BYTES_BWD
    return {"workload": runner.desc, "image_layout": runner.layout, "mode": runner.mode, "ms_per_step": round(ms, 5),
            "value": round(runner.warped_px / (ms * 1e-3) / 1e6, 1), "main_kernel_ms": round(kms, 5),
            "fwd_kernel_ms": round(float(np.mean(kt)), 5) if kt2 else None,
            "roofline_frac": round(kbytes * runner.warped_px / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "step_roofline_frac": round((BYTES_FWD + BYTES_BWD) * runner.warped_px / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def interleaved(torch, np, runner, min_time=0.2, k=25):
    """The headline step as a training loop presents it: another kernel runs between two steps.  Here the smallest such kernel, a
    torch reduction over 64 unrelated floats (the launch that ends the slow level of a loop of nothing but this step,
    profiles/r05_process_modes.txt), in front of EVERY step inside the timed blocks; `extra_kernel_ms` = what a block of those
    reductions alone costs per launch (it is part of ms_per_step)."""
    small = torch.arange(64, dtype=torch.float32, device=runner.dev)
    sink = torch.zeros((), dtype=torch.float32, device=runner.dev)

    def block(with_step):
        t0 = time.perf_counter()
        for _ in range(k):
            torch.sum(small, dim=0, out=sink)
            if with_step:
                runner.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k

    for _ in range(3):
        block(True)
    torch.cuda.synchronize()
    both, alone = [], []
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < min_time or len(both) < 5:
        both.append(block(True))
    for _ in range(5):
        alone.append(block(False))
    ms, extra = float(np.median(both)) * 1e3, float(np.median(alone)) * 1e3
    return {"workload": runner.desc, "ms_per_step": round(ms, 5), "extra_kernel_ms": round(extra, 5),
            "value": round(runner.warped_px / (ms * 1e-3) / 1e6, 1),
            "step_roofline_frac": round((BYTES_FWD + BYTES_BWD) * runner.warped_px / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "a 64-float torch reduction launched before every step, inside the timed blocks (its own time is included)"}


def link_path(torch, np, runner, min_time=0.12, k=25, use_graph=False):
    """The drop-in path a user of the reference calls: SFMLearnerLoss.__call__ from FULL-RESOLUTION frames (pyramid
    launch included) + loss.backward() (models/base_model.py:48-124), buffers cached across calls."""
    links = importlib.import_module(PKG + ".links")
    cs = importlib.import_module(PKG + ".chainer_surface")
    r = runner
    model = links.SFMLearnerLoss(dict(seq_len=r.n_src + 1, smooth_reg=r.cfg.get("smooth_reg", 0.0), exp_reg=0.0,
                                      ssim_rate=r.cfg.get("ssim_rate", 0.0)), smooth_mode=r.cfg.get("smooth_mode", "second_order"), use_graph=use_graph)
    K, disps, poses = r.common
    vd, vp = [cs.Variable(a) for a in disps], [cs.Variable(a) for a in poses]
    tgt, src = r.full

    def step():
        for v in vd + vp:
            v.cleargrad()
        loss = model(tgt, src, K, None, vd, vp)
        loss.backward()
        return loss

    for _ in range(8):
        step()
    torch.cuda.synchronize()
    blocks = []
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < min_time or len(blocks) < 3:
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / k)
    return round(float(np.median(blocks)) * 1e3, 5)


def graph_path(torch, np, runner, min_time=0.12, k=25):
    """The same step replayed from a HIP graph (one hipGraphLaunch per step instead of three kernel launches)."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            runner.step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        runner.step()
    torch.cuda.synchronize()
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
    blocks = []
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < min_time or len(blocks) < 3:
        t0 = time.perf_counter()
        for _ in range(k):
            g.replay()
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / k)
    return round(float(np.median(blocks)) * 1e3, 5)


def profile_facts(workload, layout, mode, kernel_name):
    """What the committed profiles say about the kernel this run launches (collected offline with tools/collect_profiles.sh on this
    same command; bench.py itself cannot read hardware counters): HBM-side bytes and SQ_INSTS_VALU per launch
    (profiles/<tag>_summary.json, variant workload_layout_mode), the mean issue cost of one of its vector instructions
    (profiles/<tag>_issue_model.json: the ISA of the row step priced with profiles/<tag>_op_cost_microbench.txt) and the in-kernel
    clock (profiles/<tag>_wave_stage_stamps.txt).  Missing pieces are None."""
    import re
    out = {"tag": None, "counters": None, "traffic_raw": None, "traffic_x2": None, "issue": None, "clock_ghz": None, "rocprof_avg_ns": None,
           "stale": None}
    for tag in PROFILE_TAGS:
        try:
            summ = json.load(open(os.path.join(ROOT, "profiles", "%s_summary.json" % tag)))
            var = summ["variants"]["%s_%s_%s" % (workload, layout, mode)]
            built = var["bench"].get("csrc_sha16")
            if built != csrc_sha16():
                # counters of ANOTHER build of the kernels: not this run's (round-4 verdict: figures read from files must say when
                # they are not of the tree that is running)
                out["stale"] = "profiles/%s_summary.json was collected on kernel sources %s, this run is built from %s" % (
                    tag, built or "without a recorded hash (before round 5)", csrc_sha16())
                return out
            if kernel_name not in var["kernels"] and kernel_name.replace(", false>(sfm", ">(sfm") in var["kernels"]:
                kernel_name = kernel_name.replace(", false>(sfm", ">(sfm")     # profiles of round 3: before the WARPED template argument
            if kernel_name not in var["kernels"]:        # a small launch of an L1 gradient kernel runs its three-waves-per-SIMD build
                mw = re.match(r"void sfm::loss_kernel<false, true, (\w+), false, (\d), (\w+?)(, false)?>", kernel_name)
                wide = "void sfm::loss_kernel_wide<%s, %s, %s%s>(sfm::LossArgs)" % (mw.group(1), mw.group(2), mw.group(3), mw.group(4) or "") if mw else None
                if wide in var["kernels"]:
                    kernel_name = wide
            kv = var["kernels"][kernel_name]
            out.update(tag=tag, counters=kv.get("counters_per_launch"), traffic_raw=kv.get("hbm_bytes_raw"), traffic_x2=kv.get("hbm_bytes_fetch_x2"),
                       rocprof_avg_ns=kv.get("avg_ns"))
        except Exception:
            continue
        try:
            out["issue"] = json.load(open(os.path.join(ROOT, "profiles", "%s_issue_model.json" % tag)))["kernels"][kernel_name]
        except Exception:
            pass
        try:
            m = re.search(r"in-kernel clock.*?median ([\d.]+) GHz", open(os.path.join(ROOT, "profiles", "%s_wave_stage_stamps.txt" % tag)).read())
            out["clock_ghz"] = float(m.group(1))
        except Exception:
            pass
        break
    return out


def kernel_symbol(cfg, layout, mode):
    """Name of the loss_kernel instantiation a run's dominant launch uses (as rocprofv3 prints it)."""
    expl = bool(cfg.get("exp_reg"))
    ssim = bool(cfg.get("ssim_rate")) and not expl
    smode = 0 if not cfg.get("smooth_reg") else (2 if cfg.get("smooth_mode") == "edge_aware" else 1)
    grad, loss = (True, True) if mode == "fused" else (True, False)
    tf = lambda v: "true" if v else "false"
    return "void sfm::loss_kernel<%s, %s, %s, %s, %d, %s, false>(sfm::LossArgs)" % (tf(ssim), tf(grad), tf(loss), tf(expl), smode, tf(layout == "hwc"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3_edge", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fused", choices=["separate", "fused"],
                    help="separate: sfm_loss_fwd then sfm_loss_bwd (the reference's forward / loss.backward()); "
                         "fused: one sfm_loss_fwd_bwd launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-input-warm", action="store_true", help="do not read the input arrays once before the timed region (see main)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads / paths (profiling runs)")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch of the workload (experiments only)")
    ap.add_argument("--layout", default="hwc", choices=["hwc", "planar"],
                    help="memory layout of the image pyramids resident in HBM when the timed region starts: hwc = pixel-"
                         "interleaved, as sfm_pyramid_hwc_fwd writes them (default); planar = the reference's (B,3,h,w)")
    ap.add_argument("--collective", default="step", choices=["step", "interval", "step_torch"],
                    help="under a launcher: all-reduce the five reported scalars after EVERY step with ncclAllReduce on the compute "
                         "stream (default, SURVEY.md 8(e)); once per block of --steps steps (interval); or after every step through "
                         "torch.distributed.all_reduce (step_torch)")
    ap.add_argument("--min-time", type=float, default=0.3, help="seconds of timed steps at least (blocks of --steps are repeated)")
    ap.add_argument("--max-blocks", type=int, default=400)
    ap.add_argument("--event-every", type=int, default=20, help="attach the kernel-timing events to every n-th timed step (1 = every step)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: this process only starts the ranks (fresh child processes, nothing here
        # has touched a GPU) and hands their exit code on
        sys.exit(spawn_ranks(args.gpus))

    if os.environ.get("SFM_BENCH_DRYRUN"):
        return dry_run_rank(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d, but WORLD_SIZE=%d" % (args.gpus, world))
    # SFM_BENCH_REHEARSE_ONE_GPU=1: a rehearsal of the N > 1 code path on a box with ONE GPU -- every rank runs on cuda:0 and the
    # collectives go through gloo (RCCL refuses two ranks on one device).  The line it prints says so and is not a measurement.
    rehearse = bool(os.environ.get("SFM_BENCH_REHEARSE_ONE_GPU")) and world > 1
    if rehearse:
        local = 0
    # "if the box exposes fewer devices than ranks, report the devices found": every RANK checks for itself (the parent
    # never touches a GPU; counting devices does not initialise one), says so in one line and leaves with a non-zero code
    devices_found = torch.cuda.device_count()
    if local >= devices_found:
        sys.exit("bench.py --gpus %d: rank %d (LOCAL_RANK %d) has no device -- this box exposes %d GPU(s): %d ranks cannot be measured here "
                 "(not measured, not a failure of the path)" % (args.gpus, rank, local, devices_found, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # Experiment knob (profiles/r05_process_modes.txt): SFM_BENCH_ARENA_GB=g reserves ONE g-GiB block with the caching allocator before
    # anything else is allocated and hands it back to the allocator's pool, so that every array of the run is carved out of one
    # hipMalloc (one contiguous mapping) instead of a dozen separate ones.
    if os.environ.get("SFM_BENCH_ARENA_GB"):
        arena = torch.empty((int(float(os.environ["SFM_BENCH_ARENA_GB"]) * (1 << 30)),), dtype=torch.uint8, device=dev)
        del arena
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ     # under torch.distributed.run, also at N = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ops = importlib.import_module(PKG + ".ops")
    synth = importlib.import_module(PKG + ".synth")
    ev = HipEvents()
    R = Runner(torch, np, ops, synth, dev, args.workload, args.layout, args.mode, args.batch, seed=1 + rank, norm_scale=world)
    K = max(args.steps, 1)

    # The path has no exchange on the data path (every quantity is per sample until the final means; gradients of a rank's own
    # disparities / poses never leave it).  What is exchanged is what gets REPORTED: the five scalars, summed over the shards (RCCL
    # over xGMI).  Every step writes its scalars into its own row of a device-resident log.  `--collective step` (default, SURVEY.md
    # 8(e)): ncclAllReduce(sum, fp32, 5) of the row right behind the step's launches, ON THE SAME STREAM, straight through librccl
    # (rccl.py) -- one library call, no host wait, no cross-stream event; `interval`: ONE all-reduce of the K rows at the end of the
    # block (what a trainer that reports per LogReport interval needs); `step_torch`: the per-step all-reduce through
    # torch.distributed (tools/allreduce_overhead.py, one rank: 42 us of host time per call, +8.6 us per 60 us step).
    comm, comm_note = None, None
    if use_dist and not rehearse:
        # (No multi-GPU node was available to the builder: if the direct communicator cannot be made on the node this runs on, EVERY
        #  rank falls back to torch.distributed for the collective and the line says so, rather than losing the scaling run.  The
        #  ranks AGREE on that -- rccl.connect: local steps, agreement, id broadcast, ncclCommInitRank under a deadline, agreement --
        #  so that no rank is left inside a collective the others never enter; tests/test_dist_cpu.py makes one rank fail at each step.)
        rccl = importlib.import_module(PKG + ".rccl")
        comm, comm_note = rccl.connect(rank, world, dev)
        if comm is None:
            sys.stderr.write("rank %d: no direct RCCL communicator (%s): the per-step collective goes through torch.distributed\n" % (rank, comm_note))
    # how many ranks RCCL ITSELF counts (ncclCommCount; round-5 verdict item 5): on the line as config.rccl_ranks
    rccl_ranks = None
    if comm is not None:
        rccl_ranks = comm.count()
        if rccl_ranks != world or comm.user_rank() != rank:
            sys.exit("rank %d: the RCCL communicator counts %d ranks and calls this one %d, the launcher said %d / %d" % (
                rank, rccl_ranks, comm.user_rank(), world, rank))
    raw_stream = torch.cuda.current_stream(dev).cuda_stream

    def reduce_rows(t):
        if comm is not None:
            comm.all_reduce_sum_f32(t, raw_stream)
        else:                      # the one-GPU rehearsal: gloo (RCCL refuses two ranks on one device)
            dist.all_reduce(t)

    n_log = max(K, args.warmup, 1)
    loss_log = torch.zeros((n_log, 5), dtype=torch.float32, device=dev)
    rows = [loss_log[k] for k in range(n_log)]
    # The dominant kernel is timed live, inside the timed region, by HIP events attached to its dispatch -- on every
    # `--event-every`-th step: a dispatch that carries events costs the step about 5 us (the runtime brackets it with
    # barrier packets), which is the step's business on the sampled steps only.
    ev_every = max(1, args.event_every)
    # (the sampled steps sit in the MIDDLE of their stretch of ev_every steps: the first step of a block starts on an idle GPU, behind
    #  the barrier + synchronize that opens the block, and is not representative of the steps the block is timed over)
    # Round 5: with the driver's K = 20 one sampled step per block was 0.25 us on every step of the headline.  The stretch between two
    # sampled steps is now counted ACROSS blocks and, once the number of blocks is known, widened so that the whole timed region carries
    # about EVENT_SAMPLES of them (never more often than --event-every): the kernel's mean duration rests on as many launches as it
    # needs, and the steps pay for no more.
    EVENT_SAMPLES = 64
    sample_state = {"stretch": ev_every, "count": 0, "next": min(ev_every, K) // 2}
    max_per_block = K // ev_every + 1
    event_pool = [[ev.create() for _ in range(4)] for _ in range(max_per_block)]
    events = {}          # step index within the CURRENT block -> its four events

    def plan_block_samples():
        events.clear()
        st = sample_state
        for k in range(K):
            # (never the first or the last step of a block: they border on the block's barrier + synchronize)
            if st["count"] >= st["next"] and (K <= 2 or 0 < k < K - 1) and len(events) < max_per_block:
                events[k] = event_pool[len(events)]
                st["next"] = st["count"] + st["stretch"]
            st["count"] += 1

    def run_steps(n, collective, timed):
        for k in range(n):
            R.step(out=rows[k], evs=events.get(k) if timed else None)
            if collective == "step":
                reduce_rows(rows[k])
            elif collective == "step_torch":
                dist.all_reduce(rows[k])
        if collective == "interval":
            reduce_rows(loss_log[:n])

    def timed_block(collective):
        plan_block_samples()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(K, collective, True)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed

    coll = args.collective if use_dist else None
    other_colls = [c for c in ("step", "interval", "step_torch") if c != coll]
    run_steps(args.warmup, coll, False)
    # Warm-up of the memory side, not only of the kernels (round 5, profiles/r05_process_modes.txt 1.): every input array is READ once by
    # another kernel (a sum) before the timed region.  A loop of nothing but this step freezes whatever state the memory system was left
    # in by the set-up (host-to-device copies, layout conversion); in ~40 % of the processes that state costs the step 5 % (+1 us in the
    # main kernel, +2 us behind it) for as long as nothing else runs, and reading the inputs once ends it (13 of 13 processes observed).
    # `--no-input-warm` keeps the set-up's state.
    input_warm = not args.no_input_warm
    global INPUT_WARM
    INPUT_WARM = input_warm
    # Round 6 (round-5 verdict item 6 / advisor): the level BEFORE that read is measured and put on the line too -- about twenty blocks
    # of the same K steps in the state the set-up left, `pre_warm_ms_per_step`: the figure that compares with the lines of rounds 1-4,
    # which had no such read.  (No kernel events on these blocks.)
    pre_warm = None
    if input_warm:
        events.clear()
        pw = []
        # (the SAME number of blocks on every rank -- they hold collectives -- so it follows from K alone, never from a clock: about
        #  1000 steps, between 3 and 20 blocks)
        for _ in range(min(20, max(3, 1000 // K))):
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(K, coll, False)
            torch.cuda.synchronize()
            pw.append((time.perf_counter() - t0) / K)
        pre_warm = float(np.median(pw)) * 1e3
        warm_inputs(R)
        run_steps(min(args.warmup, 5), coll, False)
    blocks, k_main, k_second = [], [], []

    def collect_kernel_times():
        k_main.extend(ev.elapsed_ms(e[0], e[1]) for e in events.values())
        if args.mode == "separate":
            k_second.extend(ev.elapsed_ms(e[2], e[3]) for e in events.values())

    blocks.append(timed_block(coll))
    collect_kernel_times()
    n_blocks = int(min(args.max_blocks, max(1, np.ceil(args.min_time / max(blocks[0], 1e-9)))))   # the same on every rank
    sample_state["stretch"] = max(ev_every, (n_blocks * K) // EVENT_SAMPLES)
    for _ in range(n_blocks - 1):
        blocks.append(timed_block(coll))
        collect_kernel_times()
    loss = loss_log[K - 1].cpu().numpy().tolist()
    per_step = np.array(blocks) / K
    ms_step = float(np.median(per_step)) * 1e3

    # N > 1: the same with the other placement of the collective, and the all-reduce of a DispNet+PoseNet-sized gradient buffer
    # (characterises xGMI; not part of this path)
    variants_ms, allreduce_probe = {}, None
    if use_dist:      # (also at N = 1 under the launcher: the same calls on a one-rank communicator)
        for oc in other_colls:
            run_steps(min(args.warmup, 3), oc, False)
            variants_ms[oc] = float(np.median([timed_block(oc) for _ in range(min(n_blocks, 5))])) / K * 1e3
    if world > 1:
        buf = torch.zeros((GRAD_BUFFER_FLOATS,), dtype=torch.float32, device=dev)
        # through the direct communicator when there is one: the bus-bandwidth figure is then RCCL's, not that of torch's host path
        big_reduce = (lambda: comm.all_reduce_sum_f32(buf, raw_stream)) if comm is not None else (lambda: dist.all_reduce(buf))
        for _ in range(3):
            big_reduce()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_ar = 10
        for _ in range(n_ar):
            big_reduce()
        torch.cuda.synchronize()
        tt = torch.tensor([(time.perf_counter() - t0) / n_ar], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_ar = float(tt.item())
        nbytes = GRAD_BUFFER_FLOATS * 4
        allreduce_probe = {"bytes": nbytes, "ms": round(t_ar * 1e3, 4), "algbw_GBs": round(nbytes / t_ar / 1e9, 1),
                           "busbw_GBs": round(2.0 * (world - 1) / world * nbytes / t_ar / 1e9, 1),
                           "xgmi_per_link_GBs": 153, "links_per_gpu": 7,
                           "issued_through": "ncclAllReduce on the direct communicator (rccl.py)" if comm is not None else "torch.distributed.all_reduce",
                           "note": "fp32 sum all-reduce of a DispNet+PoseNet-sized gradient buffer (out-of-scope trainer traffic), "
                                   "timed only to characterise RCCL over xGMI on this node"}
        del buf

    # secondary measurements (NOT part of the timed blocks above), rank 0 of a single-GPU run only
    secondary = {}
    if world == 1 and not args.no_secondary:
        def guarded(name, fn):
            try:
                secondary[name] = fn()
            except Exception as e:   # a secondary figure must never cost the headline line
                secondary[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        other_layout = "planar" if args.layout == "hwc" else "hwc"
        guarded("other_layout", lambda: quick(torch, np, ev, Runner(torch, np, ops, synth, dev, args.workload, other_layout, args.mode, args.batch)))
        guarded("other_mode", lambda: quick(torch, np, ev, Runner(torch, np, ops, synth, dev, args.workload, args.layout,
                                                                    "separate" if args.mode == "fused" else "fused", args.batch)))
        guarded("graph_ms_per_step", lambda: graph_path(torch, np, R))
        # the shape of a TRAINING loop (round-5 verdict item 6): the same step with an unrelated kernel -- a 64-float reduction, the
        # launch that ends the slow level of profiles/r05_process_modes.txt -- in front of EVERY step, inside the timed blocks
        guarded("cfg3_interleaved", lambda: interleaved(torch, np, R))
        guarded("link_ms_per_step", lambda: link_path(torch, np, R))
        for name in ("cfg3", "cfg3_edge", "cfg3_large_motion", "cfg3_smooth_disp", "cfg2", "cfg5", "cfg5_2src", "cfg5_2src_smooth_disp", "cfg1", "ref_b4"):
            if name != args.workload:
                guarded(name, lambda name=name: quick(torch, np, ev, Runner(torch, np, ops, synth, dev, name, args.layout, "fused")))
        # the drop-in link (pyramids + loss + backward, models/base_model.py:48-124) at the reference's own training batch
        guarded("ref_b4_link_ms_per_step", lambda: link_path(torch, np, Runner(torch, np, ops, synth, dev, "ref_b4", args.layout, "fused")))
        # ... and with the link's HIP-graph replay (SFMLearnerLoss(use_graph=True): pyramid + loss launches of a call whose arrays repeat
        # the previous call's addresses are replayed from one graph): what is left when the host side of the link is out of the way
        guarded("ref_b4_link_graph_ms_per_step", lambda: link_path(torch, np, Runner(torch, np, ops, synth, dev, "ref_b4", args.layout, "fused"), use_graph=True))
        # north_star's backward "scatters dL/d(depth, pose, src_img)": the same step with the OPTIONAL d_src output bound (the reference
        # discards it in training, base_model.py:71-72 `.data`; NULL is the default) -- 12 global float atomics per warped pixel
        guarded("cfg3_d_src", lambda: quick(torch, np, ev, Runner(torch, np, ops, synth, dev, args.workload, args.layout, "fused", args.batch, want_d_src=True)))
        # ... and on the smooth disparity field, where the taps of a row of samples stay inside the wave's LDS accumulation window
        guarded("cfg3_d_src_smooth_disp", lambda: quick(torch, np, ev, Runner(torch, np, ops, synth, dev, "cfg3_smooth_disp", args.layout, "fused", want_d_src=True)))

    for e4 in event_pool:
        for e in e4:
            ev.destroy(e)

    if rank == 0:
        value = R.warped_px * world / (ms_step * 1e-3) / 1e6
        if args.mode == "fused":
            kname, kbytes, kt = "loss_kernel<grad+loss> (sfm_loss_fwd_bwd)", BYTES_FWD + BYTES_BWD, k_main
        else:
            kname, kbytes, kt = "loss_kernel<grad> (sfm_loss_bwd)", BYTES_BWD, k_second
        kms = float(np.mean(kt))
        achieved = kbytes * R.warped_px / (kms * 1e-3) / 1e9
        # HBM-side bytes and issued vector instructions per launch of that kernel from the rocprofv3 PMC passes committed under profiles/
        ksym = kernel_symbol(R.cfg, args.layout, args.mode)
        facts = profile_facts(args.workload, args.layout, args.mode, ksym) if args.batch == 0 else profile_facts("-", "-", "-", "-")
        traffic = round(facts["traffic_x2"]) if facts["traffic_x2"] else None
        traffic_detail = None
        if traffic:
            traffic_detail = {"bytes_per_launch_raw": round(facts["traffic_raw"]), "bytes_per_launch_fetch_x2": traffic,
                              "source": "profiles/%s_summary.json, variant %s_%s_%s (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; KiB "
                                        "units; FETCH_SIZE doubled per MI355X_MICROARCH.md, uncalibrated for 4-24 B/lane loads)" % (
                                            facts["tag"], args.workload, args.layout, args.mode)}
        valu = float(facts["counters"]["SQ_INSTS_VALU"]) if facts["counters"] and "SQ_INSTS_VALU" in facts["counters"] else None
        roofline = {
            # the roofline the fraction is taken against: HBM (ALGORITHMIC bytes / kernel time vs the 8 TB/s line; a gather /
            # reduce path, no MFMA).  What actually limits the kernel is vector-instruction issue at its occupancy:
            # `binding_resource`, priced in `roofline_valu`.
            "bound": "hbm", "binding_resource": "valu_issue", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_detail": traffic_detail,
            "measured_hbm_frac": round(traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            "kernel": kname, "kernel_ms": round(kms, 5), "kernel_ms_median": round(float(np.median(kt)), 5),
            "kernel_ms_p10_p90": [round(float(np.percentile(kt, 10)), 5), round(float(np.percentile(kt, 90)), 5)],
            "bytes_per_warped_px": kbytes, "launches_timed": len(kt)}
        roofline["kernel_symbol"] = ksym
        if facts["stale"]:
            roofline["traffic_detail"] = {"traffic_is_null_because": facts["stale"]}
        if facts["rocprof_avg_ns"]:
            roofline["rocprof_kernel_ms_in_profiles"] = round(facts["rocprof_avg_ns"] * 1e-6, 5)
        roofline_valu = None
        if valu and facts["issue"] and facts["clock_ghz"]:
            # issue-bound floor of a launch: counted vector instructions x the mean issue cost of one of them at this kernel's
            # occupancy (ISA of the row step priced with the measured per-class costs), spread over the SIMDs, at the in-kernel clock
            iss = facts["issue"]
            floor_ms = valu * iss["mean_issue_cycles_per_valu"] / N_SIMD / (facts["clock_ghz"] * 1e9) * 1e3
            roofline_valu = {"bound": "valu_issue", "valu_insts_per_launch": valu, "waves_per_simd": iss["waves_per_simd"],
                             "mean_issue_cycles_per_valu": iss["mean_issue_cycles_per_valu"], "clock_GHz": facts["clock_ghz"],
                             "floor_ms": round(floor_ms, 5), "frac": round(floor_ms / kms, 4),
                             "source": "SQ_INSTS_VALU: profiles/%s_summary.json; cycles per instruction: profiles/%s_issue_model.json (ISA of HEAD "
                                       "priced with profiles/%s_op_cost_microbench.txt); clock: profiles/%s_wave_stage_stamps.txt" % ((facts["tag"],) * 4)}
        out = {
            "metric": "warp+photo-loss fwd+bwd Mpixels/s @128x416x4scales; % HBM roofline",
            "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "devices_found": devices_found, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": R.desc, "per_gpu_batch": R.B, "global_batch": R.B * world, "H": R.H, "W": R.W, "n_src": R.n_src,
                       "n_scales": R.n_scales, "mode": args.mode, "image_layout": args.layout, "warped_px_per_gpu_step": R.warped_px,
                       "input_warm_read": input_warm,
                       "rccl_ranks": rccl_ranks,
                       "collective": collective_line(coll, comm is not None, rehearse, comm_note) if use_dist else None,
                       "parallelism": ("batch-sharded x%d, no exchange on the data path; RCCL all-reduce of the 5 reported scalars: %s" % (
                           world, COLLECTIVE_NOTES[coll])) if use_dist else "single GPU, no collective"},
            "timing": {"blocks": len(blocks), "steps_per_block": K, "timed_s": round(float(np.sum(blocks)), 4),
                       "ms_per_step_median": round(ms_step, 5), "ms_per_step_p10": round(float(np.percentile(per_step, 10)) * 1e3, 5),
                       "ms_per_step_p90": round(float(np.percentile(per_step, 90)) * 1e3, 5),
                       "ms_per_step_first_block": round(float(per_step[0]) * 1e3, 5), "ms_per_step_min": round(float(per_step.min()) * 1e3, 5),
                       # the blocks in time order, thinned to at most 48 entries: how the box's state moved while it was measured
                       "ms_per_step_series": [round(float(v) * 1e3, 4) for v in per_step[::max(1, len(per_step) // 48)]]},
            "roofline": roofline,
            "roofline_valu": roofline_valu,
            "roofline_valu_null_because": facts["stale"] if roofline_valu is None else None,
            "step_roofline_frac": round((BYTES_FWD + BYTES_BWD) * R.warped_px / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            # the same blocks BEFORE the one-off read of the inputs (config.input_warm_read): the protocol of rounds 1-4
            "pre_warm_ms_per_step": round(pre_warm, 5) if pre_warm else None,
            "pre_warm_step_roofline_frac": round((BYTES_FWD + BYTES_BWD) * R.warped_px / (pre_warm * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pre_warm else None,
            "csrc_sha16": csrc_sha16(),
            "loss5": [round(v, 6) for v in loss],
        }
        for oc, ms in variants_ms.items():
            out[VARIANT_KEYS[oc]] = {"ms_per_step": round(ms, 5), "value": round(R.warped_px * world / (ms * 1e-3) / 1e6, 1),
                                     "note": COLLECTIVE_NOTES[oc]}
        if world > 1:
            out["allreduce_160MB"] = allreduce_probe
        if rehearse:
            out["rehearsal"] = ("SFM_BENCH_REHEARSE_ONE_GPU: all %d ranks ran on ONE GPU and the collectives went through gloo -- this line "
                                "exercises the N > 1 code path, it is NOT a measurement" % world)
            out["config"]["parallelism"] = "REHEARSAL on one GPU (gloo): " + out["config"]["parallelism"]
        out.update(secondary)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.destroy()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
