"""Direct RCCL binding for the ONE collective of the sharded search: an
all-gather of 10 KB per rank and step.  torch.distributed's wrapper costs ~25 us of host time
per call (work object, event record / wait, group start / end) -- more than the scan of a
125 k-row shard -- so the strong-scaled job is host-bound from 4 GPUs on.  Calling
ncclAllGather through ctypes on the SAME librccl.so torch ships (no second copy of the
library) brings the step to three plain enqueues on the lane's own HIP stream.

The communicator is this module's own: rank 0 draws an ncclUniqueId, torch.distributed (any
backend) broadcasts its 128 bytes, every rank calls ncclCommInitRank.  One process = one GPU."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char, c_char_p, c_int, c_size_t, c_void_p

NCCL_INT64 = 4   # ncclDataType_t: int8 0, uint8 1, int32 2, uint32 3, int64 4, ...


class NcclUniqueId(Structure):
    _fields_ = [("internal", c_char * 128)]


def _load():
    import torch
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = ctypes.CDLL(path)
    lib.ncclGetUniqueId.restype = c_int
    lib.ncclGetUniqueId.argtypes = [POINTER(NcclUniqueId)]
    lib.ncclCommInitRank.restype = c_int
    lib.ncclCommInitRank.argtypes = [POINTER(c_void_p), c_int, NcclUniqueId, c_int]
    lib.ncclAllGather.restype = c_int
    lib.ncclAllGather.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p]
    lib.ncclCommDestroy.restype = c_int
    lib.ncclCommDestroy.argtypes = [c_void_p]
    lib.ncclGetErrorString.restype = c_char_p
    lib.ncclGetErrorString.argtypes = [c_int]
    return lib


class RcclComm:
    """all_gather_i64(send_ptr, recv_ptr, count, stream_ptr): enqueue on the given HIP stream."""

    def __init__(self, rank: int, world: int, device, group=None):
        import torch
        import torch.distributed as dist
        self.lib = _load()
        self.rank, self.world = rank, world
        uid = NcclUniqueId()
        if rank == 0:
            self._check(self.lib.ncclGetUniqueId(byref(uid)), "ncclGetUniqueId")
        if world > 1:
            # 128 bytes through the existing process group (GPU tensor for nccl, host tensor otherwise)
            on_gpu = dist.get_backend(group) == "nccl"
            t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
            t = t.to(device) if on_gpu else t
            dist.broadcast(t, src=0, group=group)
            ctypes.memmove(byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
        self.comm = c_void_p()
        with torch.cuda.device(device):
            self._check(self.lib.ncclCommInitRank(byref(self.comm), world, uid, rank), "ncclCommInitRank")

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.ncclGetErrorString(rc).decode()} ({rc})")

    def all_gather_i64(self, send_ptr: int, recv_ptr: int, count: int, stream_ptr):
        rc = self.lib.ncclAllGather(c_void_p(send_ptr), c_void_p(recv_ptr), count, NCCL_INT64, self.comm, stream_ptr)
        if rc != 0:
            self._check(rc, "ncclAllGather")

    def destroy(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = c_void_p()
