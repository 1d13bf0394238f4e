#!/usr/bin/env python3
"""Cost of the per-step collective of bench.py's N > 1 path, measured on ONE GPU (world size 1, RCCL):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 tools/allreduce_overhead.py
Prints ms per step of: the steps alone; steps + one async all-reduce of the 5 scalars per step (what bench.py times);
the host time of the all_reduce call by itself; and the same collective issued through a pre-captured HIP graph if the
stack allows capturing it."""
import importlib, os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bench = importlib.import_module("bench")
PKG = "sfm-learner-chainer_amd"
ops = importlib.import_module(PKG + ".ops"); synth = importlib.import_module(PKG + ".synth")
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
dist.init_process_group("nccl", rank=int(os.environ.get("RANK", 0)), world_size=int(os.environ.get("WORLD_SIZE", 1)), device_id=dev)
R = bench.Runner(torch, np, ops, synth, dev, "cfg3_edge", "hwc", "fused", 0, seed=1, norm_scale=dist.get_world_size())
K = 200
log = torch.zeros((K, 5), dtype=torch.float32, device=dev)
rows = [log[k] for k in range(K)]


def timed(fn, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / K * 1e3)
    return float(np.median(out))


def plain():
    for k in range(K):
        R.step(out=rows[k])


def per_step():
    works = []
    for k in range(K):
        R.step(out=rows[k])
        works.append(dist.all_reduce(rows[k], async_op=True))
    for w in works:
        w.wait()


def host_only():
    t0 = time.perf_counter()
    works = [dist.all_reduce(rows[k], async_op=True) for k in range(K)]
    t = (time.perf_counter() - t0) / K * 1e6
    for w in works:
        w.wait()
    return t


def per_step_sync():
    for k in range(K):
        R.step(out=rows[k])
        dist.all_reduce(rows[k])


side = torch.cuda.Stream()


def per_step_side():
    cur = torch.cuda.current_stream()
    works = []
    for k in range(K):
        R.step(out=rows[k])
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            works.append(dist.all_reduce(rows[k], async_op=True))
    for w in works:
        w.wait()
    cur.wait_stream(side)


for _ in range(2):
    plain(); per_step(); per_step_sync(); per_step_side()
print("steps + blocking all-reduce every step %.4f ms/step" % timed(per_step_sync))
print("steps + all-reduce issued from a side stream %.4f ms/step" % timed(per_step_side))
print("steps alone                         %.4f ms/step" % timed(plain))
print("steps + async all-reduce every step %.4f ms/step" % timed(per_step))
torch.cuda.synchronize()
print("host time of one all_reduce call    %.1f us" % host_only())
t0 = time.perf_counter()
for k in range(K):
    R.step(out=rows[k])
host_step = (time.perf_counter() - t0) / K * 1e6
torch.cuda.synchronize()
print("host time of one step's launches    %.1f us" % host_step)
try:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        R.step(out=rows[0]); dist.all_reduce(rows[0])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        R.step(out=rows[0])
        dist.all_reduce(rows[0])
    torch.cuda.synchronize()

    def graphed():
        for k in range(K):
            g.replay()
    graphed()
    print("step + all-reduce replayed from a HIP graph %.4f ms/step" % timed(graphed))
except Exception as e:
    print("graph capture of the collective not available here: %s: %s" % (type(e).__name__, str(e).splitlines()[0][:160]))
dist.destroy_process_group()
