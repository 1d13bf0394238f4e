"""The N > 1 path on CPU: two processes over gloo.  Each rank takes its contiguous batch shard,
evaluates the loss with norm_B = global batch (here with the oracle standing in for the HIP
kernels, which need a GPU) and the ranks all-reduce the five reported scalars exactly as
bench.py / the GPU path do through `dist.allreduce_losses`."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["total_loss", "pixel_loss", "smooth_loss", "exp_loss", "ssim_loss"]
CFG = dict(smooth_reg=0.1, ssim_rate=0.15)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dmod = importlib.import_module("sfm-learner-chainer_amd.dist")
    synth = importlib.import_module("sfm-learner-chainer_amd.synth")
    from oracle import sfm_oracle as O
    r, w, device = dmod.init(backend="gloo")
    assert (r, w) == (rank, world) and dmod.world() == world and dmod.rank() == rank
    full = synth.make_inputs(B=5, H=16, W=24, n_src=2, n_scales=2, seed=4)      # 5 samples over 2 ranks: 3 + 2
    sh = dmod.shard_inputs(full)
    assert sh["global_B"] == 5 and sh["B"] == (3 if rank == 0 else 2)
    res = O.sfm_loss(sh["tgt_pyr"], sh["src_pyr"], sh["intrinsics"], sh["disps"], sh["poses"], backward=True,
                     norm_batch=sh["global_B"], **CFG)
    loss5 = torch.tensor([res[k] for k in KEYS], dtype=torch.float64)
    # bench.py keeps one row of scalars per step and reduces the rows of a reporting interval with ONE collective
    log = torch.stack([loss5 * (k + 1) for k in range(3)])      # three "steps" of an interval
    dmod.allreduce_losses(loss5)
    dmod.allreduce_losses(log)
    # the hand-over of RCCL's unique id (rccl.Communicator: made on rank 0, broadcast over the process group that is already up):
    # 128 bytes that include zeros and values above 127 must arrive unchanged on every rank
    rccl = importlib.import_module("sfm-learner-chainer_amd.rccl")
    uid = bytes((37 * k + 11) % 256 for k in range(rccl.NCCL_UNIQUE_ID_BYTES))
    got = rccl._broadcast_bytes(uid if rank == 0 else bytes(rccl.NCCL_UNIQUE_ID_BYTES), device)
    assert got == uid and len(got) == rccl.NCCL_UNIQUE_ID_BYTES
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), loss5=loss5.numpy(), log=log.numpy(), d_disp0=res.d_disps[0],
             d_pose0=res.d_poses[0])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_batch_sharding_over_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    synth = importlib.import_module("sfm-learner-chainer_amd.synth")
    dmod = importlib.import_module("sfm-learner-chainer_amd.dist")
    from oracle import sfm_oracle as O
    full = synth.make_inputs(B=5, H=16, W=24, n_src=2, n_scales=2, seed=4)
    ref = O.sfm_loss(full["tgt_pyr"], full["src_pyr"], full["intrinsics"], full["disps"], full["poses"], backward=True, **CFG)
    want = np.array([ref[k] for k in KEYS])
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for o in outs:                                    # every rank holds the global scalars after the all-reduce
        np.testing.assert_allclose(o["loss5"], want, rtol=1e-6)
        np.testing.assert_allclose(o["log"], np.outer([1.0, 2.0, 3.0], want), rtol=1e-6)   # one collective per interval
    # a rank's gradients are those of its samples in the full batch: no exchange needed
    for r, o in enumerate(outs):
        lo, hi = dmod.shard_range(5, r, world)
        np.testing.assert_allclose(o["d_disp0"], ref.d_disps[0][lo:hi], rtol=0, atol=1e-5 * np.abs(ref.d_disps[0]).max())
        np.testing.assert_allclose(o["d_pose0"], ref.d_poses[0][lo:hi], rtol=0, atol=1e-5 * np.abs(ref.d_poses[0]).max())
