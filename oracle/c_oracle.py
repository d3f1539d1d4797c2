"""ctypes wrapper of oracle/search_oracle.c.  TEST INFRASTRUCTURE ONLY (same
rules as oracle/search.py).  Build with `make -C oracle` (done by
__graft_entry__.build())."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsearch_oracle.so")
_lib = None


def build() -> str:
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_search.restype = ctypes.c_int
        _lib.oracle_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def search(q16: np.ndarray, c16: np.ndarray, k: int, id_base: int = 0):
    lib = _load()
    q = np.ascontiguousarray(q16, dtype=np.float16)
    c = np.ascontiguousarray(c16, dtype=np.float16)
    B, D = q.shape
    N = c.shape[0]
    out_s = np.empty((B, k), dtype=np.float64)
    out_i = np.empty((B, k), dtype=np.int64)
    rc = lib.oracle_search(q.ctypes.data, c.ctypes.data, B, N, D, k, id_base,
                           out_s.ctypes.data, out_i.ctypes.data)
    if rc != 0:
        raise MemoryError("oracle_search failed")
    return out_s, out_i
