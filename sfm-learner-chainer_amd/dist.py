"""Batch-sharded multi-GPU use of the loss path: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU hosts for tests).

The path shards over samples: every quantity is per-sample until the final
means (models/base_model.py:109,111,115,184-185), so each rank runs the fused loss on its own
samples with `norm_B` = the GLOBAL batch.  Gradients w.r.t. a rank's own disparities / poses
need no exchange; the only collective of the path is the sum of the five reported scalars.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

__all__ = ["init", "shard_range", "shard_inputs", "allreduce_losses", "world", "rank"]


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init(backend=None):
    """Initialises the default process group from the torchrun environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world, device)."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    r = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local)
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if w > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if use_gpu else "gloo")
        if backend == "nccl":
            dist.init_process_group(backend, rank=r, world_size=w, device_id=device)
        else:
            dist.init_process_group(backend, rank=r, world_size=w)
    return r, w, device


def shard_range(global_batch, r=None, w=None):
    """Contiguous [lo, hi) slice of the batch owned by rank r of w (sizes differ by at most one)."""
    r = rank() if r is None else r
    w = world() if w is None else w
    base, rem = divmod(global_batch, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def shard_inputs(inputs, r=None, w=None):
    """Slices every per-sample array of a `synth.make_inputs`-style dict to this rank's samples."""
    lo, hi = shard_range(inputs["B"], r, w)
    sl = slice(lo, hi)
    out = dict(inputs)
    for key in ("tgt", "src", "intrinsics"):
        out[key] = inputs[key][sl]
    for key in ("tgt_pyr", "src_pyr", "disps", "poses"):
        out[key] = [a[sl] for a in inputs[key]]
    out["masks"] = [a[sl] for a in inputs["masks"]] if inputs.get("masks") is not None else None
    out["B"] = hi - lo
    out["global_B"] = inputs["B"]
    return out


def allreduce_losses(loss5):
    """Sum of the per-shard shares of (total, pixel, smooth, exp, ssim) over all ranks, in place."""
    if world() > 1:
        dist.all_reduce(loss5, op=dist.ReduceOp.SUM)
    return loss5
