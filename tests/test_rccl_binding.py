"""CPU check of the direct RCCL binding (rag_fin_amd/rccl.py): the librccl.so that torch ships
loads, exports the five entry points the sharded step uses, and the ctypes signatures are the
ones ncclAllGather / ncclCommInitRank expect.  No communicator is created (that needs a GPU;
the one-rank path runs in `RAGFIN_FORCE_SHARDED=1 python bench.py` on the GPU box)."""
import ctypes

import pytest


def test_librccl_exports_and_signatures():
    pytest.importorskip("torch")
    from rag_fin_amd import rccl
    try:
        lib = rccl._load()
    except OSError as e:
        pytest.skip(f"librccl.so not loadable here: {e}")
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllGather", "ncclCommDestroy", "ncclGetErrorString"):
        assert hasattr(lib, name), name
    assert ctypes.sizeof(rccl.NcclUniqueId) == 128          # NCCL_UNIQUE_ID_BYTES
    assert lib.ncclAllGather.argtypes[2] is ctypes.c_size_t and lib.ncclAllGather.argtypes[3] is ctypes.c_int
    assert lib.ncclCommInitRank.argtypes[2] is rccl.NcclUniqueId   # passed BY VALUE
    assert rccl.NCCL_INT64 == 4
    assert lib.ncclGetErrorString(0) is not None
