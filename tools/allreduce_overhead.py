#!/usr/bin/env python3
"""Run under torch.distributed.run (any N): cost of the per-step all-reduce of the five scalars, sync vs async."""
import importlib, sys, os, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("sfm-learner-chainer_amd.ops"); synth = importlib.import_module("sfm-learner-chainer_amd.synth")
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local); dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
d = synth.make_inputs(B=32, H=128, W=416, n_src=2, n_scales=4, seed=1 + rank)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fl = ops.FusedLoss(smooth_reg=0.1, ssim_rate=0.15).bind([t(a) for a in d["tgt_pyr"]], [t(a) for a in d["src_pyr"]], t(d["intrinsics"]), [t(a) for a in d["disps"]], [t(a) for a in d["poses"]], norm_B=32 * world)
slots = [torch.zeros(5, device=dev) for _ in range(4)]
def run(mode, n=200):
    works = [None] * 4
    for it in range(n + 20):
        if it == 20:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        k = it % 4
        if works[k] is not None: works[k].wait(); works[k] = None
        fl.forward_backward(out=slots[k])
        if mode == "sync": dist.all_reduce(slots[k])
        elif mode == "async": works[k] = dist.all_reduce(slots[k], async_op=True)
        elif mode == "every8" and it % 8 == 7: dist.all_reduce(slots[k])
    t_issue = (time.perf_counter() - t0) / n
    for w in works:
        if w is not None: w.wait()
    torch.cuda.synchronize()
    return t_issue * 1e6, (time.perf_counter() - t0) / n * 1e6
for mode in ("none", "sync", "async", "every8", "none"):
    ti, tw = run(mode)
    if rank == 0: print("%-7s host issue %.1f us/step, wall %.1f us/step" % (mode, ti, tw))
dist.destroy_process_group()
