"""The one collective of the path, issued straight through RCCL: ``ncclAllReduce`` (sum, fp32) of the five reported
scalars ON THE STREAM THE LOSS KERNELS RUN ON ("one ncclAllReduce(sum, fp32, count=5) per step over
RCCL/xGMI").  ctypes over the librccl.so instance that torch has already loaded -- the copy that shares torch's HIP runtime, so
that torch's streams and device pointers are valid handles for it (a second copy from /opt/rocm would bring a second runtime).

Why not ``torch.distributed.all_reduce``: it is 42 us of host time and two cross-stream events around a 4 us kernel
(tools/allreduce_overhead.py), i.e. +8.6 us on a 60 us step; an ``ncclAllReduce`` enqueued on the compute stream is one
library call, and the reduced row is ready in stream order -- nothing waits on the host.

Reference mechanism being replaced: the stock Chainer updaters chosen by YAML (config_utils.py:122-133,156-161), whose NCCL
traffic (parameter gradients) belongs to the out-of-scope trainer; the loss path itself only ever exchanges what it REPORTS
(models/base_model.py:119-123).
"""
from __future__ import annotations

import ctypes as C

import torch

NCCL_UNIQUE_ID_BYTES = 128          # rccl.h
NCCL_FLOAT32, NCCL_SUM = 7, 0       # ncclDataType_t, ncclRedOp_t (rccl.h)


class _UniqueId(C.Structure):
    # (bytes, not c_char: ctypes hands a c_char array back truncated at its first NUL, and an id is full of them)
    _fields_ = [("internal", C.c_ubyte * NCCL_UNIQUE_ID_BYTES)]


def _id_bytes(uid):
    return C.string_at(C.byref(uid), NCCL_UNIQUE_ID_BYTES)


class RcclError(RuntimeError):
    pass


def _loaded_librccl():
    """Path of the librccl.so this process has mapped (torch links against its own copy), or None."""
    with open("/proc/self/maps") as f:
        for line in f:
            if "librccl.so" in line:
                return line.split()[-1]
    return None


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _loaded_librccl()
        if path is None:
            # torch maps librccl with libtorch_hip; reaching this means a torch build without RCCL
            raise RcclError("librccl.so is not mapped into this process (torch built without RCCL?)")
        L = C.CDLL(path)
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetErrorString.argtypes = [C.c_int]
        L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RcclError("%s failed: %s (ncclResult_t %d)" % (what, lib().ncclGetErrorString(rc).decode(), rc))


def _broadcast_bytes(payload, device):
    """Rank 0's 128 bytes to every rank over the process group that is already up (whatever its backend)."""
    import torch.distributed as dist
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
    if dist.get_backend() == "nccl":
        t = t.to(device)
    dist.broadcast(t, src=0)
    return bytes(t.cpu().numpy().tobytes())


class Communicator:
    """One RCCL communicator per process (= per GPU).  `rank`, `world` as torch.distributed reports them; the unique id is
    made on rank 0 and handed round by `broadcast` (default: a torch.distributed broadcast on the existing process group)."""

    def __init__(self, rank, world, device, broadcast=None):
        L = lib()
        self.rank, self.world, self.device = rank, world, device
        uid = _UniqueId()
        if rank == 0:
            _check(L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if world > 1:
            raw = (broadcast or (lambda b: _broadcast_bytes(b, device)))(_id_bytes(uid) if rank == 0 else bytes(NCCL_UNIQUE_ID_BYTES))
            if len(raw) != NCCL_UNIQUE_ID_BYTES:
                raise RcclError("the unique id arrived with %d bytes instead of %d" % (len(raw), NCCL_UNIQUE_ID_BYTES))
            C.memmove(C.byref(uid), raw, NCCL_UNIQUE_ID_BYTES)
        self._comm = C.c_void_p()
        with torch.cuda.device(device):
            _check(L.ncclCommInitRank(C.byref(self._comm), world, uid, rank), "ncclCommInitRank")

    def all_reduce_sum_f32(self, tensor, stream=None):
        """In place, asynchronous, in stream order on `stream` (default: torch's current stream of the tensor's device)."""
        if tensor.dtype != torch.float32 or not tensor.is_cuda or not tensor.is_contiguous():
            raise TypeError("all_reduce_sum_f32: a contiguous float32 device tensor is required")
        st = stream if stream is not None else torch.cuda.current_stream(tensor.device).cuda_stream
        p = C.c_void_p(tensor.data_ptr())
        _check(lib().ncclAllReduce(p, p, tensor.numel(), NCCL_FLOAT32, NCCL_SUM, self._comm, C.c_void_p(st)), "ncclAllReduce")
        return tensor

    def destroy(self):
        if self._comm:
            lib().ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()
