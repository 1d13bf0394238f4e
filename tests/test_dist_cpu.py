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


def _worker8(rank, world, port, out_dir):
    """BASELINE cfg4's rank count on the CPU: B = 19 over eight ranks (shards of 3, 3, 3, 2, 2, 2, 2, 2), norm_B global."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dmod = importlib.import_module("sfm-learner-chainer_amd.dist")
    synth = importlib.import_module("sfm-learner-chainer_amd.synth")
    rccl = importlib.import_module("sfm-learner-chainer_amd.rccl")
    from oracle import sfm_oracle as O
    r, w, device = dmod.init(backend="gloo")
    full = synth.make_inputs(B=19, H=16, W=24, n_src=2, n_scales=2, seed=9)
    sh = dmod.shard_inputs(full)
    lo, hi = dmod.shard_range(19)
    assert sh["global_B"] == 19 and sh["B"] == hi - lo == (3 if rank < 3 else 2)
    res = O.sfm_loss(sh["tgt_pyr"], sh["src_pyr"], sh["intrinsics"], sh["disps"], sh["poses"], backward=True, norm_batch=19, **CFG)
    loss5 = torch.tensor([res[k] for k in KEYS], dtype=torch.float64)
    dmod.allreduce_losses(loss5)
    uid = bytes((37 * k + 11) % 256 for k in range(rccl.NCCL_UNIQUE_ID_BYTES))
    got = rccl._broadcast_bytes(uid if rank == 0 else bytes(rccl.NCCL_UNIQUE_ID_BYTES), device)
    assert got == uid
    # the agreed protocol with a stand-in library: eight communicators, each counting eight ranks
    log = []
    comm, note = rccl.connect(rank, world, device, init_timeout_s=30.0, library=lambda: _FakeRccl(rank, "ok", log, world))
    assert comm is not None and note is None and comm.count() == world and comm.user_rank() == rank
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), loss5=loss5.numpy(), lo=lo, hi=hi, d_pose0=res.d_poses[0])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_eight_rank_batch_sharding_over_gloo(tmp_path):
    """Round-5 verdict item 5: cfg4 is EIGHT ranks, and no test ran more than two.  B = 19 global: the remainders of shard_range
    (3, 3, 3, 2, 2, 2, 2, 2), norm_B global on every rank, the shard scalars add up to the single-process oracle to 1e-6, a rank's
    d_pose is that of its samples in the full batch, the 128-byte id arrives intact on all eight, rccl.connect ends with a
    communicator on all eight whose own rank count (ncclCommCount) is eight."""
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    synth = importlib.import_module("sfm-learner-chainer_amd.synth")
    dmod = importlib.import_module("sfm-learner-chainer_amd.dist")
    from oracle import sfm_oracle as O
    full = synth.make_inputs(B=19, H=16, W=24, n_src=2, n_scales=2, seed=9)
    ref = O.sfm_loss(full["tgt_pyr"], full["src_pyr"], full["intrinsics"], full["disps"], full["poses"], backward=True, **CFG)
    want = np.array([ref[k] for k in KEYS])
    covered = []
    for r in range(world):
        o = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        np.testing.assert_allclose(o["loss5"], want, rtol=1e-6)
        lo, hi = int(o["lo"]), int(o["hi"])
        assert (lo, hi) == dmod.shard_range(19, r, world)
        covered += list(range(lo, hi))
        np.testing.assert_allclose(o["d_pose0"], ref.d_poses[0][lo:hi], rtol=0, atol=1e-5 * np.abs(ref.d_poses[0]).max())
    assert covered == list(range(19))                       # every sample on exactly one rank, in order


# ------------------------------------------------------------------------------------------------
# rccl.connect: the agreement protocol around the direct communicator, with one rank made to fail at each step
# ------------------------------------------------------------------------------------------------
class _FakeRccl:
    """Stands in for the bound librccl on a host without GPUs: the four entry points rccl.connect uses, with a failure injected on
    rank 1 according to `mode`."""

    def __init__(self, rank, mode, log, world=2):
        self.rank, self.mode, self.log, self.world = rank, mode, log, world

    def ncclCommCount(self, comm, p_n):
        p_n._obj.value = self.world
        return 0

    def ncclCommUserRank(self, comm, p_r):
        p_r._obj.value = self.rank
        return 0

    def ncclGetUniqueId(self, p_uid):
        for k in range(128):
            p_uid._obj.internal[k] = (7 * k + 3) % 256
        return 0

    def ncclCommInitRank(self, p_comm, world, uid, rank):
        import time
        got = bytes(uid.internal)
        assert got == bytes((7 * k + 3) % 256 for k in range(128)), "the unique id did not arrive intact"
        if self.rank == 1 and self.mode == "init_error":
            return 5
        if self.rank == 1 and self.mode == "init_hang":
            time.sleep(20)
        if self.rank == 1 and self.mode == "init_late":          # returns a LIVE communicator after the caller's deadline
            time.sleep(5)
        p_comm._obj.value = 0x1000 + self.rank
        self.log.append("init")
        return 0

    def ncclGetErrorString(self, rc):
        return b"invalid usage (injected)"

    def ncclCommAbort(self, comm):
        self.log.append("abort")
        return 0

    def ncclCommDestroy(self, comm):
        self.log.append("destroy")
        return 0


def _connect_worker(rank, world, port, out_dir, mode):
    import json
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    rccl = importlib.import_module("sfm-learner-chainer_amd.rccl")
    bench = importlib.import_module("bench")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []

    def library():
        if rank == 1 and mode == "local_fail":
            raise rccl.RcclError("librccl.so is not mapped into this process (injected)")
        return _FakeRccl(rank, mode, log)

    if rank == 1 and mode == "short_id":                         # the broadcast hands this rank fewer bytes than an id has
        full_bcast = rccl._broadcast_bytes
        rccl._broadcast_bytes = lambda payload, device: full_bcast(payload, device)[:100]
    comm, note = rccl.connect(rank, world, torch.device("cpu"), init_timeout_s=3.0, library=library)
    if mode == "init_late":
        import time
        time.sleep(4.0)                                          # let rank 1's abandoned call come back
    # whatever happened, the process group is still in step: the next collective matches on every rank
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    line = bench.collective_line("step", comm is not None, False, note)
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(dict(has_comm=comm is not None, note=note, log=log, sum=float(t.item()), line=line), f)
    if comm is not None:
        comm.destroy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["ok", "local_fail", "init_error", "init_hang", "init_late", "short_id"])
def test_direct_communicator_failure_on_one_rank_ends_in_an_agreed_fallback(tmp_path, mode):
    """Round-4 advisor finding + verdict item 6(a): rank 1 fails to bind the library / gets an error from ncclCommInitRank / never
    returns from it.  BOTH ranks must end without a direct communicator (the per-step collective then goes through
    torch.distributed), rank 0 must have given up the communicator it did get, the line must say FALLBACK, and the process group
    must still be in step.  With no failure both ranks get their communicator."""
    import json
    world = 2
    mp.spawn(_connect_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    outs = [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % r))) for r in range(world)]
    for o in outs:
        assert o["sum"] == 3.0                                   # the collective after connect() matched on both ranks
    if mode == "ok":
        assert all(o["has_comm"] and o["note"] is None for o in outs)
        assert all("FALLBACK" not in o["line"] for o in outs)
        return
    assert not any(o["has_comm"] for o in outs), outs
    assert all(o["note"] for o in outs)
    assert all("FALLBACK" in o["line"] for o in outs)
    if mode == "local_fail":
        assert outs[0]["log"] == [] and outs[1]["log"] == []        # nobody entered ncclCommInitRank: no rank was left waiting in it
        assert "injected" in outs[1]["note"] and "another rank" in outs[0]["note"]
    elif mode == "short_id":
        # (round-5 advisor finding) a short id is a vote of 0, never a zero-padded id: rank 1 does not even call ncclCommInitRank
        assert outs[0]["log"] == ["init", "abort"] and outs[1]["log"] == []
        assert "100 bytes" in outs[1]["note"]
    elif mode == "init_late":
        # (round-5 advisor finding) the call rank 1 gave up on came back with a live communicator: the helper thread aborted it itself
        assert outs[0]["log"] == ["init", "abort"] and outs[1]["log"] == ["init", "abort"]
        assert "did not return" in outs[1]["note"]
    else:
        assert outs[0]["log"] == ["init", "abort"]                  # rank 0 had its communicator and gave it up
        assert "abort" not in outs[1]["log"]
        assert ("did not return" in outs[1]["note"]) if mode == "init_hang" else ("ncclCommInitRank failed" in outs[1]["note"])
